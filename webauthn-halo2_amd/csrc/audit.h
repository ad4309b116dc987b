// audit.h — ZK_OPT_STREAM_AUDIT: a happens-before ledger of the context's streams (debug option, off by default).
//
// Why: a context runs up to four streams (main, tail, transform, MSM) and every hand-off between them is a hand-written
// event pair.  Round 5's first transform stream missed one — two streams' NTTs shared the ping-pong scratch without an order —
// and zk_prove returned ZK_OK with wrong bytes for about one proof in 1 500 (found by a soak, not by a test).  With the option
// on, every enqueue the engine makes names the buffers it reads and writes; the ledger keeps, per buffer, the stream and
// logical time of its last writer and of its last reader per stream, and, per stream, the vector clock of what that stream is
// ordered after (its own enqueues, the events it waited on, what the host had synchronised with when the enqueue was made).
// An enqueue that reads a buffer whose last writer its stream is not ordered after (RAW), or writes one with an unordered
// earlier reader or writer (WAR / WAW), is a violation: counted, the first one described, and the entry point that was running
// returns ZK_EINTERNAL.  The check is on the ORDER the program establishes, not on timing: the round-5 bug shows on the
// first proof that takes the faulty path, not once in 1 500.
//
// Granularity: a buffer is identified by its base pointer as the engine passes it around (a zk_poly's vector, a workspace
// member, an MSM lane's workspace, the NTT scratch, a pinned result buffer); sub-ranges of one vector are the same buffer.
// Coverage: the engine-level operators (MSM passes with their head / tail streams, NTT batches, evaluations, the quotient) and
// every launch of the single prover (csrc/prover.hip); the lock-step prover's merged launches run on the main stream only and
// are covered through the engine-level operators they call.  Cost when off: one predictable branch per enqueue.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <initializer_list>
#include <string>
#include <unordered_map>
#include <vector>

namespace zk {

struct StreamAudit {
    static constexpr int NS = 4;  // main, tail, transform, MSM stream of the context
    bool on = false;
    hipStream_t streams[NS] = {nullptr, nullptr, nullptr, nullptr};  // index 0 = the main stream; the others as they are created
    uint64_t clk[NS] = {0, 0, 0, 0};      // enqueues made on the stream so far
    uint64_t seen[NS][NS] = {};           // seen[s][t]: everything up to enqueue seen[s][t] of stream t is ordered before s's next enqueue
    uint64_t host[NS] = {0, 0, 0, 0};     // the host has waited for enqueue host[t] of stream t (and what that was ordered after)
    struct Stamp {
        int s;
        uint64_t at;
        uint64_t seen[NS];
    };
    std::unordered_map<hipEvent_t, Stamp> events;
    struct Buf {
        int ws = -1;  // last writer's stream
        uint64_t wat = 0;
        const char* wsite = "";
        uint64_t rat[NS] = {0, 0, 0, 0};  // last read per stream
        const char* rsite[NS] = {"", "", "", ""};
    };
    std::unordered_map<const void*, Buf> bufs;
    std::unordered_map<const void*, const void*> base_of;  // pointer -> base of its device allocation (cache; cleared per proof)
    uint64_t violations = 0, checks = 0;
    std::string first;

    // a buffer's identity: the base of the DEVICE allocation the pointer lies in (h_ext + i n and h_ext are ONE buffer: the
    // ledger is conservative about sub-ranges of device vectors); anything else — a block of a pinned staging ring (whose blocks
    // are reused one by one while uploads of the others are in flight), a pinned result buffer, the address of an MSM workspace
    // record standing for the workspace — is its own identity
    const void* key(const void* p) {
        auto it = base_of.find(p);
        if (it != base_of.end()) return it->second;
        const void* k = p;
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, p) == hipSuccess && at.type == hipMemoryTypeDevice) {
            hipDeviceptr_t base = nullptr;
            size_t size = 0;
            if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)p) == hipSuccess && base) k = base;
        }
        (void)hipGetLastError();  // (a host heap pointer is an error for both calls: not ours to report)
        base_of.emplace(p, k);
        return k;
    }

    int sid(hipStream_t st) {
        for (int i = 0; i < NS; i++)
            if (streams[i] == st) return i;
        for (int i = 1; i < NS; i++)
            if (!streams[i]) {
                streams[i] = st;
                return i;
            }
        return 0;  // (more streams than the engine has: treated as the main stream)
    }
    void reset() {
        for (int i = 0; i < NS; i++) {
            clk[i] = host[i] = 0;
            for (int j = 0; j < NS; j++) seen[i][j] = 0;
        }
        events.clear();
        bufs.clear();
        base_of.clear();
        violations = checks = 0;
        first.clear();
    }
    // what stream s is ordered after right now: its own past, its waits, and everything the host has synchronised with
    bool ordered_after(int s, int t, uint64_t at) const { return s == t || seen[s][t] >= at || host[t] >= at; }
    void violate(const char* kind, const char* site, const void* p, int s, int t, uint64_t at, const char* other) {
        violations++;
        if (first.empty()) {
            char msg[512];
            snprintf(msg, sizeof(msg), "%s hazard at '%s' (stream %d) on buffer %p: stream %d's enqueue #%llu ('%s') is not ordered before it",
                     kind, site, s, p, t, (unsigned long long)at, other);
            first = msg;
        }
    }
    // one enqueue on `st`: the buffers it reads and the buffers it writes (a buffer it updates in place goes into both)
    void op(hipStream_t st, std::initializer_list<const void*> reads, std::initializer_list<const void*> writes, const char* site) {
        if (!on) return;
        op_v(st, reads.begin(), reads.size(), writes.begin(), writes.size(), site);
    }
    void op_v(hipStream_t st, const void* const* reads, size_t nr, const void* const* writes, size_t nw, const char* site) {
        if (!on) return;
        const int s = sid(st);
        const uint64_t at = ++clk[s];
        for (size_t i = 0; i < nr; i++) {
            if (!reads[i]) continue;
            const void* p = key(reads[i]);
            Buf& b = bufs[p];
            checks++;
            if (b.ws >= 0 && !ordered_after(s, b.ws, b.wat)) violate("read-after-write", site, p, s, b.ws, b.wat, b.wsite);
            b.rat[s] = at;
            b.rsite[s] = site;
        }
        for (size_t i = 0; i < nw; i++) {
            if (!writes[i]) continue;
            const void* p = key(writes[i]);
            Buf& b = bufs[p];
            checks++;
            if (b.ws >= 0 && !ordered_after(s, b.ws, b.wat)) violate("write-after-write", site, p, s, b.ws, b.wat, b.wsite);
            for (int t = 0; t < NS; t++)
                if (b.rat[t] && !ordered_after(s, t, b.rat[t])) violate("write-after-read", site, p, s, t, b.rat[t], b.rsite[t]);
            b.ws = s;
            b.wat = at;
            b.wsite = site;
        }
    }
    void record(hipEvent_t ev, hipStream_t st) {
        if (!on) return;
        const int s = sid(st);
        Stamp x;
        x.s = s;
        x.at = clk[s];
        for (int t = 0; t < NS; t++) x.seen[t] = seen[s][t];
        events[ev] = x;
    }
    void wait(hipStream_t st, hipEvent_t ev) {
        if (!on) return;
        const int s = sid(st);
        auto it = events.find(ev);
        if (it == events.end()) return;  // never recorded: the wait is a no-op for HIP as well
        const Stamp& x = it->second;
        for (int t = 0; t < NS; t++)
            if (x.seen[t] > seen[s][t]) seen[s][t] = x.seen[t];
        if (x.at > seen[s][x.s]) seen[s][x.s] = x.at;
    }
    // the host has waited for the event / the stream / the device
    void host_event(hipEvent_t ev) {
        if (!on) return;
        auto it = events.find(ev);
        if (it == events.end()) return;
        const Stamp& x = it->second;
        for (int t = 0; t < NS; t++)
            if (x.seen[t] > host[t]) host[t] = x.seen[t];
        if (x.at > host[x.s]) host[x.s] = x.at;
    }
    void host_stream(hipStream_t st) {
        if (!on) return;
        const int s = sid(st);
        for (int t = 0; t < NS; t++)
            if (seen[s][t] > host[t]) host[t] = seen[s][t];
        if (clk[s] > host[s]) host[s] = clk[s];
    }
    void host_all() {
        if (!on) return;
        for (int t = 0; t < NS; t++) host[t] = clk[t];
    }
    // the host reads a buffer a stream wrote (a pinned result buffer): it must have waited for the writer
    void host_read(const void* p, const char* site) {
        if (!on || !p) return;
        auto it = bufs.find(key(p));
        if (it == bufs.end()) return;
        checks++;
        const Buf& b = it->second;
        if (b.ws >= 0 && host[b.ws] < b.wat) violate("host-read-before-wait", site, p, -1, b.ws, b.wat, b.wsite);
    }
    // the host overwrites a buffer streams read or wrote (pinned staging): every such enqueue must be complete
    void host_write(const void* p, const char* site) {
        if (!on || !p) return;
        auto it = bufs.find(key(p));
        if (it == bufs.end()) return;
        checks++;
        const Buf& b = it->second;
        if (b.ws >= 0 && host[b.ws] < b.wat) violate("host-write-before-wait", site, p, -1, b.ws, b.wat, b.wsite);
        for (int t = 0; t < NS; t++)
            if (b.rat[t] && host[t] < b.rat[t]) violate("host-write-before-read-done", site, p, -1, t, b.rat[t], b.rsite[t]);
    }
};

}  // namespace zk
