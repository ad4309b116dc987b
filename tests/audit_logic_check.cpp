// Host-only check of the stream ledger's logic (csrc/audit.h): scripted enqueue sequences on made-up stream / event handles —
// no device is touched (StreamAudit::key falls back to the pointer itself when HIP knows nothing about it).
#include <cstdio>
#include "audit.h"
using namespace zk;

static int fails = 0;
#define EXPECT(cond)                                           \
    do {                                                       \
        if (!(cond)) {                                         \
            printf("FAILED line %d: %s\n", __LINE__, #cond);   \
            fails++;                                           \
        }                                                      \
    } while (0)

int main() {
    hipStream_t M = (hipStream_t)0x10, T = (hipStream_t)0x20, X = (hipStream_t)0x30, S = (hipStream_t)0x40;
    hipEvent_t e1 = (hipEvent_t)0x100, e2 = (hipEvent_t)0x200, e3 = (hipEvent_t)0x300;
    int a, b, scratch, host_buf;  // four "buffers"
    {   // off: nothing is recorded, nothing is refused
        StreamAudit A;
        A.op(M, {}, {&a}, "w");
        A.op(X, {&a}, {}, "r");
        EXPECT(A.violations == 0 && A.checks == 0);
    }
    {   // read-after-write across streams: refused without an event, accepted with record / wait
        StreamAudit A;
        A.on = true;
        A.streams[0] = M;
        A.op(M, {}, {&a}, "produce a");
        A.op(X, {&a}, {&b}, "consume a (unordered)");
        EXPECT(A.violations == 1 && A.first.find("read-after-write") != std::string::npos);
        A.reset();
        A.streams[0] = M;
        A.op(M, {}, {&a}, "produce a");
        A.record(e1, M);
        A.wait(X, e1);
        A.op(X, {&a}, {&b}, "consume a");
        EXPECT(A.violations == 0);
        // ... but the NEXT write of a on M must wait for X's read (write-after-read)
        A.op(M, {}, {&a}, "overwrite a too early");
        EXPECT(A.violations == 1 && A.first.find("write-after-read") != std::string::npos);
    }
    {   // transitivity: X waits on T's event, T had waited on M's
        StreamAudit A;
        A.on = true;
        A.streams[0] = M;
        A.op(M, {}, {&a}, "produce a");
        A.record(e1, M);
        A.wait(T, e1);
        A.op(T, {&a}, {&b}, "a -> b");
        A.record(e2, T);
        A.wait(X, e2);
        A.op(X, {&a, &b}, {}, "reads both");
        EXPECT(A.violations == 0);
        // an event recorded BEFORE the producing enqueue orders nothing
        A.record(e3, M);
        A.op(M, {}, {&scratch}, "late producer");
        A.wait(S, e3);
        A.op(S, {&scratch}, {}, "stale wait");
        EXPECT(A.violations == 1);
    }
    {   // round 5's bug: two streams ping-pong through ONE scratch; the second waits on an event taken before the first's transform
        StreamAudit A;
        A.on = true;
        A.streams[0] = M;
        A.op(M, {}, {&a, &b}, "blinding rows");
        A.record(e1, M);                              // ev_rows
        A.op(M, {&a, &scratch}, {&a, &scratch}, "NTT batch");  // transforms on the main stream
        A.wait(X, e1);                                // the side stream waits for the rows only
        A.op(X, {&b, &scratch}, {&b, &scratch}, "NTT batch");
        EXPECT(A.violations >= 1 && A.first.find("NTT batch") != std::string::npos);
    }
    {   // host synchronisation orders everything enqueued afterwards, on any stream; host reads need the wait
        StreamAudit A;
        A.on = true;
        A.streams[0] = M;
        A.op(T, {}, {&host_buf}, "tail writes the pinned sums");
        A.record(e1, T);
        A.host_read(&host_buf, "host reads too early");
        EXPECT(A.violations == 1);
        A.reset();
        A.streams[0] = M;
        A.op(T, {}, {&host_buf}, "tail writes the pinned sums");
        A.record(e1, T);
        A.host_event(e1);
        A.host_read(&host_buf, "host reads after the wait");
        A.op(M, {}, {&host_buf}, "next pass's tail on another stream");  // ordered: the host had waited before enqueueing
        EXPECT(A.violations == 0);
        A.op(X, {}, {&a}, "x writes a");
        A.host_stream(X);
        A.op(S, {&a}, {}, "s reads a after the host synchronised x");
        EXPECT(A.violations == 0);
        A.host_write(&a, "host overwrites a while s may still read it");
        EXPECT(A.violations == 1);
    }
    printf("audit logic: %d failures\n", fails);
    return fails ? 1 : 0;
}
