// engine.hip — context, device memory and the C ABI of libzkmi355.so
// (declarations and the reference routines each entry point replaces:
// include/zkmi355.h).
#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <new>

#include "ctx.h"

namespace zk {
// poly.hip
uint32_t eval_blocks(uint32_t n);
void launch_eval(const Fr* c, uint32_t n, const Fr& x, Fr* scratch, hipStream_t st);
void launch_srs_lagrange_scalars(const Fr* tw, uint32_t n, const Fr& s, const Fr& c, Fr* out, hipStream_t st);
void launch_srs_fixed_base(const Fr* scalars, uint32_t n, const G1Affine* table, G1Affine* out, hipStream_t st);

G1Affine g1_jac_to_affine_host(const G1Jac& p) {
    G1Affine r;
    if (p.z.is_zero()) {
        r.x = Fq::zero();
        r.y = Fq::zero();
        return r;
    }
    const Fq zi = fe_inv_fast(p.z);
    const Fq zi2 = fe_sqr(zi);
    r.x = fe_mul(p.x, zi2);
    r.y = fe_mul(p.y, fe_mul(zi2, zi));
    return r;
}
}  // namespace zk

int ctx_bind(zk_ctx* c) {
    hipError_t e = hipSetDevice(c->device);
    if (e != hipSuccess) {
        c->last_hip = (int)e;
        return ZK_EHIP;
    }
    return ZK_OK;
}

int ctx_ensure_scratch(zk_ctx* c, size_t n) {
    if (c->scratch_n >= n) return ZK_OK;
    if (c->xform_stream) aud_sync(c, c->xform_stream);  // (transforms in flight use the buffer)
    if (c->scratch) hipFree(c->scratch);
    c->scratch = nullptr;
    c->scratch_n = 0;
    hipError_t e = hipMalloc(&c->scratch, n * sizeof(Fr));
    if (e != hipSuccess) {
        c->last_hip = (int)e;
        return ZK_ENOMEM;
    }
    c->scratch_n = n;
    return ZK_OK;
}

int ctx_get_twiddles(zk_ctx* c, uint32_t log_n, const Fr** out) {
    auto it = c->twiddles.find(log_n);
    if (it != c->twiddles.end()) {
        *out = it->second;
        return ZK_OK;
    }
    if (log_n > 28) return ZK_EINVAL;
    Fr* tw = nullptr;
    const size_t n = (size_t)1 << log_n;
    if (hipMalloc(&tw, n * sizeof(Fr)) != hipSuccess) return ZK_ENOMEM;
    launch_twiddles(tw, fr_omega(log_n), (uint32_t)n, c->stream);
    aud_sync(c, c->stream);  // made once per context and size; read from any of the context's streams afterwards
    c->twiddles[log_n] = tw;
    *out = tw;
    return ZK_OK;
}

int ctx_get_twiddles_ntt(zk_ctx* c, uint32_t log_n, const Fr** out) {
    auto it = c->twiddles_ntt.find(log_n);
    if (it != c->twiddles_ntt.end()) {
        *out = it->second;
        return ZK_OK;
    }
    if (log_n > 28) return ZK_EINVAL;
    Fr* tw = nullptr;
    const size_t n = (size_t)1 << log_n;
    if (hipMalloc(&tw, n * sizeof(Fr)) != hipSuccess) return ZK_ENOMEM;
    launch_twiddles_internal(tw, fr_omega(log_n), (uint32_t)n, c->stream);
    aud_sync(c, c->stream);  // made once per context and size; read from any of the context's streams afterwards
    c->twiddles_ntt[log_n] = tw;
    *out = tw;
    return ZK_OK;
}

int ctx_get_twiddles_ninv(zk_ctx* c, uint32_t log_n, const Fr** out) {
    auto it = c->twiddles_ninv.find(log_n);
    if (it != c->twiddles_ninv.end()) {
        *out = it->second;
        return ZK_OK;
    }
    if (log_n > 28) return ZK_EINVAL;
    Fr* tw = nullptr;
    const size_t n = (size_t)1 << log_n;
    if (hipMalloc(&tw, n * sizeof(Fr)) != hipSuccess) return ZK_ENOMEM;
    launch_twiddles_scaled(tw, fr_omega(log_n), fe_inv(fr_from_u64(n)), (uint32_t)n, c->stream);
    aud_sync(c, c->stream);  // made once per context and size; read from any of the context's streams afterwards
    c->twiddles_ninv[log_n] = tw;
    *out = tw;
    return ZK_OK;
}

namespace zk {
void launch_scale(Fr* a, const Fr& c, uint32_t n, hipStream_t st);
}
int ctx_get_coset_points(zk_ctx* c, uint32_t log_n, const Fr** out) {
    auto it = c->coset_points.find(log_n);
    if (it != c->coset_points.end()) {
        *out = it->second;
        return ZK_OK;
    }
    const Fr* tw = nullptr;
    int rc = ctx_get_twiddles(c, log_n, &tw);
    if (rc) return rc;
    const size_t n = (size_t)1 << log_n;
    Fr* xs = nullptr;
    if (hipMalloc(&xs, n * sizeof(Fr)) != hipSuccess) return ZK_ENOMEM;
    if (hipMemcpyAsync(xs, tw, n * sizeof(Fr), hipMemcpyDeviceToDevice, c->stream) != hipSuccess) {
        hipFree(xs);
        return ZK_EHIP;
    }
    launch_scale(xs, c->zeta, (uint32_t)n, c->stream);
    c->coset_points[log_n] = xs;
    *out = xs;
    return ZK_OK;
}

// columns per fixed-base launch.  Batching makes the accumulate launch bigger (fuller waves: -15 % per column
// already at two columns of 2^19) and replaces several reduction tails by one longer one; measured best
// (whole proofs): 2 at 2^19, growing as the columns get shorter and launch overheads dominate
static uint32_t batch_for(const zk_ctx* c, size_t n) {
    size_t b = ((size_t)1 << 20) / (n ? n : 1);
    if (c->opt_msm_batch) b = c->opt_msm_batch;  // zk_ctx_set_option(ZK_OPT_MSM_BATCH)
    if (b < 1) b = 1;
    if (b > MSM_MAX_BATCH) b = MSM_MAX_BATCH;
    return (uint32_t)b;
}

uint32_t ctx_msm_max_batch(const zk_ctx* c) {
    if (c->srs_k < 0 || !c->table_c) return 1;
    return batch_for(c, (size_t)1 << c->srs_k);
}

// device staging buffer `which` of at least `bytes` (kept for the next call: a Rust host patched at best_multiexp /
// best_fft calls the seam a dozen times per proof with the same sizes)
static int seam_buffer(zk_ctx* c, int which, size_t bytes, void** out) {
    if (c->seam_bytes[which] < bytes) {
        if (c->seam_buf[which]) hipFree(c->seam_buf[which]);
        c->seam_buf[which] = nullptr;
        c->seam_bytes[which] = 0;
        hipError_t e = hipMalloc(&c->seam_buf[which], bytes);
        if (e != hipSuccess) {
            c->last_hip = (int)e;
            return ZK_ENOMEM;
        }
        c->seam_bytes[which] = bytes;
    }
    *out = c->seam_buf[which];
    return ZK_OK;
}

// `table_window` != 0: the workspace serves the fixed-base mode over the resident SRS (window = the tables'); otherwise
// arbitrary bases, whose windows stop at 15 bits
static int get_msm_ws(zk_ctx* c, int lane, size_t n, uint32_t table_window, MsmWorkspace** out) {
    size_t want = 1;
    while (want < n) want <<= 1;
    if (want < 1024) want = 1024;
    const uint32_t cw = table_window ? table_window : msm_auto_window_generic(want);
    zk_ctx::MsmLane& L = c->lanes[lane];
    MsmWorkspace*& slot = table_window ? L.ws : L.ws_gen;
    // columns per pass the workspace must take: the single prover's batches, or a lock-step batch's wider passes
    const uint32_t cols = table_window ? std::max(batch_for(c, want), std::min<uint32_t>(c->msm_min_cols, MSM_MAX_BATCH)) : 1u;
    if (slot && (msm_ws_max_n(slot) != want || msm_ws_window(slot) != cw || msm_ws_max_batch(slot) < cols)) {
        aud_sync(c, c->stream);  // (a pass of the lane that has been collected may still have kernels of its tail queued behind others)
        if (c->tail_stream) aud_sync(c, c->tail_stream);
        msm_workspace_destroy(slot);
        slot = nullptr;
    }
    if (!slot) {
        hipError_t e;
        slot = msm_workspace_create(want, cw, &e, cols);
        if (!slot && e != hipErrorInvalidValue && !c->poly_spare.empty()) {
            ctx_release_spares(c);
            (void)hipGetLastError();
            slot = msm_workspace_create(want, cw, &e, cols);
        }
        if (!slot) {
            c->last_hip = (int)e;
            return e == hipErrorInvalidValue ? ZK_EINVAL : ZK_ENOMEM;
        }
    }
    L.ws_run = slot;
    *out = slot;
    return ZK_OK;
}

// ---- streams are kept, not destroyed, and made in a deliberate order (round 6).  The HIP runtime ties a stream to one of its
// (four) hardware queues when the stream is made: the first four streams of a process get a queue each, every later one the queue
// with the fewest streams on it (ties: the highest queue) — read off rocprofv3's Queue_Id with tools/queue_map.py.  Streams that
// share a queue run in order, so WHICH streams share matters: a process that had destroyed a set of contexts got 190 instead of 228
// proofs/s from its next four pipelines (k = 17, tools/inflight_k17.py 4 4: main streams sharing queues), and a context whose lone
// proof finds its tail and its transform stream on one queue takes 12.0 instead of 11.1 ms (k = 19).  So the first context of a
// device makes, in this order, the MAIN streams of the device's first four contexts (queues 0 .. 3) and then four blocks of four
// side streams (each block: queues 3, 2, 1, 0) - and the same again for four more contexts (main streams on queues 3 .. 0).  Context slot i owns main stream i and, from block i, the side streams that do
// not sit on its main's queue: its tail stream on queue 3 - i (so that the tails of the first two pipelines do not meet either),
// its transform and MSM streams on the other two.  zk_ctx_destroy drains the slot's streams and frees the slot for the next
// context; contexts beyond the eight slots (and a main stream made at its own priority, ZK_OPT_STREAM_PRIORITY) make their streams as
// before and destroy them.  The pool is never freed (40 idle streams per device for the life of the process).  If another runtime
// assigns queues differently nothing breaks: this is placement, not correctness.
namespace {
constexpr int POOL_SLOTS = 8;  // two layers of four: slots 4 .. 7 repeat the pattern (their main streams land on queues 3 .. 0)
struct StreamSlots {
    bool primed = false;
    hipStream_t main[POOL_SLOTS] = {};
    hipStream_t side[POOL_SLOTS][4] = {};  // [slot][j]: block `slot`, j-th made: queue 3 - j
    bool used[POOL_SLOTS] = {};
};
struct StreamPool {
    std::mutex mu;
    std::map<int, StreamSlots> dev;
};
StreamPool& stream_pool() {
    static StreamPool* p = new StreamPool();  // (leaked on purpose: the HIP runtime may be gone before static destructors run)
    return *p;
}
// the hardware queue of slot s's main stream under the runtime's rule (first four streams: a queue each; then the least loaded
// queue, ties to the highest): layer 0 = queues 0 .. 3, layer 1 (made after layer 0's side blocks) = queues 3 .. 0
constexpr int slot_main_queue(int s) { return s < 4 ? s : 7 - s; }
}  // namespace
// a free slot of the device (current device = `device`), or -1: the caller makes its own streams
static int pool_take_slot(int device, hipStream_t* main_out) {
    StreamPool& p = stream_pool();
    std::lock_guard<std::mutex> lk(p.mu);
    StreamSlots& d = p.dev[device];
    if (!d.primed) {
        d.primed = true;
        bool ok = true;
        for (int layer = 0; layer < POOL_SLOTS / 4 && ok; layer++) {
            for (int i = 4 * layer; i < 4 * layer + 4 && ok; i++) ok = hipStreamCreate(&d.main[i]) == hipSuccess;
            for (int i = 4 * layer; i < 4 * layer + 4 && ok; i++)
                for (int j = 0; j < 4 && ok; j++) ok = hipStreamCreate(&d.side[i][j]) == hipSuccess;
        }
        if (!ok) {  // (out of resources: no slots on this device, contexts make their own streams)
            for (int i = 0; i < POOL_SLOTS; i++) d.used[i] = true;
        }
    }
    // layer 1 is handed out from the top: slot 7's main stream shares queue 0 with slot 0's, so the fifth context doubles up with the
    // FIRST one (the oldest, most likely idle: a set-up or probe context) rather than with the fourth
    static const int order[POOL_SLOTS] = {0, 1, 2, 3, 7, 6, 5, 4};
    for (int k = 0; k < POOL_SLOTS; k++) {
        const int i = order[k];
        if (!d.used[i]) {
            d.used[i] = true;
            *main_out = d.main[i];
            return i;
        }
    }
    return -1;
}
static void pool_release_slot(int device, int slot) {
    StreamPool& p = stream_pool();
    std::lock_guard<std::mutex> lk(p.mu);
    StreamSlots& d = p.dev[device];
    hipStreamSynchronize(d.main[slot]);
    for (int j = 0; j < 4; j++) hipStreamSynchronize(d.side[slot][j]);
    d.used[slot] = false;
}
// role: 0 the tail stream, 1 the transform stream, 2 the MSM stream
int ctx_side_stream(zk_ctx* c, hipStream_t* out, int role) {
    if (*out) return ZK_OK;
    if (c->stream_slot >= 0) {
        const int i = c->stream_slot, mq = slot_main_queue(i);
        int js[3], m = 0;
        js[m++] = mq;  // side j sits on queue 3 - j: the tail takes the queue opposite the main's (3 - mq, never mq itself)
        for (int j = 0; j < 4; j++)
            if (j != mq && j != 3 - mq) js[m++] = j;  // (j = 3 - mq sits on the main's queue: the block's spare)
        StreamPool& p = stream_pool();
        std::lock_guard<std::mutex> lk(p.mu);
        *out = p.dev[c->device].side[i][js[role]];
        return ZK_OK;
    }
    return hipStreamCreate(out) == hipSuccess ? ZK_OK : ZK_EHIP;
}

// the two further streams a LONE proof spreads over (ctx.h xform_stream, msm_stream), made when the first such proof asks
int ctx_lone_streams(zk_ctx* c) {
    if (ctx_side_stream(c, &c->xform_stream, 1) || ctx_side_stream(c, &c->msm_stream, 2)) return ZK_EHIP;
    return ZK_OK;
}

// ---- how busy the device is, as far as this process can see: contexts that enqueued an MSM pass within the last few
// milliseconds.  Every context owns a slot of its device's table and stamps it in ctx_msm_begin_batch — the one place every
// MSM pass goes through, whoever asked for it (zk_prove, zk_commit / zk_commit_batch, zk_msm_srs, zk_msm_bn254, zk_keygen,
// zk_pk_read): a host that drives the phase-level ABI from four threads is seen exactly like four zk_prove calls (round 4
// counted zk_prove calls only).  PROCESS-LOCAL: contexts of other processes on the same GPU are invisible.
namespace {
constexpr int ACT_DEVICES = 64, ACT_SLOTS = 64;
constexpr int64_t ACT_WINDOW_NS = 4 * 1000 * 1000;  // a proving context enqueues a pass every 0.3 .. 1.5 ms
// ... but not during its quotient / evaluation / multi-open phases, which under four pipelines last longer than the window: a
// context inside a whole-proof call (zk_prove, zk_prove_batch) holds its slot "active" for the length of the call
// (ctx_activity_hold), or the count would dip to two or three several times per proof and passes of the OTHER contexts would take
// the side-stream regime under full load
constexpr int64_t ACT_HELD = INT64_MAX;
std::atomic<int64_t> g_act_ts[ACT_DEVICES][ACT_SLOTS];
std::atomic<uint64_t> g_act_used[ACT_DEVICES];
int64_t act_now() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace
void ctx_activity_register(zk_ctx* c) {
    c->act_slot = -1;
    if (c->device < 0 || c->device >= ACT_DEVICES) return;
    std::atomic<uint64_t>& used = g_act_used[c->device];
    uint64_t cur = used.load();
    for (;;) {
        if (~cur == 0) return;  // more than 64 contexts on one device: the surplus ones are not counted
        const int slot = __builtin_ctzll(~cur);
        if (used.compare_exchange_weak(cur, cur | (1ull << slot))) {
            g_act_ts[c->device][slot].store(0);
            c->act_slot = slot;
            return;
        }
    }
}
void ctx_activity_unregister(zk_ctx* c) {
    if (c->act_slot < 0 || c->device < 0 || c->device >= ACT_DEVICES) return;
    g_act_ts[c->device][c->act_slot].store(0);
    g_act_used[c->device].fetch_and(~(1ull << c->act_slot));
    c->act_slot = -1;
}
// stamps this context and returns the number of contexts (this one included) active on its device
int ctx_activity_touch(zk_ctx* c) {
    if (c->device < 0 || c->device >= ACT_DEVICES) return 1;
    const int64_t now = act_now();
    if (c->act_slot >= 0 && !c->act_held) g_act_ts[c->device][c->act_slot].store(now);
    int active = c->act_slot >= 0 ? 0 : 1;
    uint64_t used = g_act_used[c->device].load();
    while (used) {
        const int slot = __builtin_ctzll(used);
        used &= used - 1;
        const int64_t ts = g_act_ts[c->device][slot].load();
        if (ts && (ts == ACT_HELD || now - ts < ACT_WINDOW_NS)) active++;
    }
    return active;
}
// a whole-proof call begins / ends on this context (prover.hip ProveQuiesce)
void ctx_activity_hold(zk_ctx* c, bool on) {
    if (on && c->opt_no_activity_hold) return;  // ZK_OPT_ACTIVITY_HOLD = 1: round 5's rule (stamps only)
    if (!on && c->opt_activity_pinned) return;  // ZK_OPT_ACTIVITY_HOLD = 2: the host holds this context active itself
    c->act_held = on;
    if (c->act_slot < 0 || c->device < 0 || c->device >= ACT_DEVICES) return;
    g_act_ts[c->device][c->act_slot].store(on ? ACT_HELD : act_now());
}

int ctx_msm_begin_batch(zk_ctx* c, int lane, const Fr* const* d_scalars, uint32_t batch, const G1Affine* d_bases, size_t n) {
    if (lane < 0 || lane >= zk_ctx::MSM_LANES || c->lanes[lane].busy || batch == 0) return ZK_EINVAL;
    zk_ctx::MsmLane& L = c->lanes[lane];
    MsmWorkspace* ws;
    // commits against the resident SRS use the precomputed window tables
    const G1Affine* table = nullptr;
    uint32_t stride = 0;
    bool ident = true;  // arbitrary bases: the accumulation tests every operand
    if (c->srs_k >= 0 && c->table_c) {
        if (d_bases == c->g) {
            table = c->g_table;
            ident = c->g_has_identity;
        } else if (d_bases == c->g_lagrange) {
            table = c->g_lagrange_table;
            ident = c->g_lagrange_has_identity;
        }
        stride = 1u << c->srs_k;
    }
    int rc = get_msm_ws(c, lane, table ? (size_t)stride : n, table ? c->table_c : 0u, &ws);
    if (rc) return rc;
    if (batch > 1 && (!table || batch > msm_ws_max_batch(ws))) return ZK_EINVAL;
    // where this pass's reduction tail runs (ctx.h tail_stream): the side stream for up to two proofs in flight on the device
    const int active = ctx_activity_touch(c);
    const uint32_t above = c->opt_tail_main_above ? c->opt_tail_main_above : 2u;  // ZK_OPT_MSM_TAIL_MAIN_ABOVE; measured default (ctx.h)
    const bool tail_on_main = c->opt_tail_stream == 2 || (c->opt_tail_stream == 0 && (uint32_t)active > above);
    if ((!tail_on_main || c->msm_side) && !c->tail_stream && ctx_side_stream(c, &c->tail_stream, 0)) return ZK_EHIP;  // made on first use (see zk_ctx_create)
    L.tail = tail_on_main ? c->stream : c->tail_stream;
    if (L.tail == c->stream) c->acc_n[ZK_T_MSM_TAIL_MAIN]++;
    hipStream_t hs = c->stream;  // where the pass's head and accumulation run
    if (c->msm_side && c->msm_stream) {
        hs = c->msm_stream;
        HIPCHK(c, aud_record(c, c->ev_msm_in, c->stream));  // everything the main stream holds so far: the pass's inputs among it
        HIPCHK(c, aud_wait(c, hs, c->ev_msm_in));
        if (L.tail == c->stream) L.tail = c->tail_stream;    // (never a tail behind the main stream's later kernels)
    }
    HIPCHK(c, aud_record(c, L.t_head[0], hs));
    msm_ws_set_t1_mode(ws, c->opt_msm_t1);
    HIPCHK(c, msm_run(ws, d_scalars, batch, d_bases, n, hs, L.host_buf, &L.nwin, &L.cw, L.t_acc, table, stride, L.tail,
                      L.head_done, !table || ident));
    if (c->audit.on) {
        // the ledger's twin of what msm_run enqueued: head + accumulation on hs (reads the columns, fills the lane's workspace),
        // the head_done hand-off, the tail on L.tail (reads the workspace, writes the lane's pinned result buffer)
        const void* rd[MSM_MAX_BATCH];
        for (uint32_t q = 0; q < batch; q++) rd[q] = d_scalars[q];
        const void* wr[1] = {ws};
        c->audit.op_v(hs, rd, batch, wr, 1, "MSM pass: sort head + accumulation");
        if (L.tail != hs) {
            c->audit.record(L.head_done, hs);
            c->audit.wait(L.tail, L.head_done);
        }
        c->audit.op(L.tail, {ws}, {ws, L.host_buf}, "MSM pass: reduction tail");
    }
    HIPCHK(c, aud_record(c, L.tail_done, L.tail));
    HIPCHK(c, aud_record(c, L.t_head[1], hs));
    c->msm_launches++;
    L.n = n;
    L.batch = batch;
    L.table = table;
    L.fixed = table != nullptr;
    L.busy = true;
    return ZK_OK;
}

int ctx_msm_begin(zk_ctx* c, int lane, const Fr* d_scalars, const G1Affine* d_bases, size_t n) {
    return ctx_msm_begin_batch(c, lane, &d_scalars, 1, d_bases, n);
}

int ctx_msm_end_batch(zk_ctx* c, int lane, G1Jac* out) {
    if (lane < 0 || lane >= zk_ctx::MSM_LANES || !c->lanes[lane].busy) return ZK_EINVAL;
    zk_ctx::MsmLane& L = c->lanes[lane];
    L.busy = false;
    HIPCHK(c, aud_esync(c, L.tail_done));
    c->audit.host_read(L.host_buf, "MSM pass: the host collects the sums");
    if (L.fixed && L.n > 0 && msm_wide_redo_count(L.ws_run, L.host_buf, L.batch)) {
        // a degenerate basis (equal or opposite points): the unchecked accumulation reported lanes to redo with the checked loop
        HIPCHK(c, msm_wide_redo(L.ws_run, L.batch, L.n, L.tail, L.host_buf, L.table));
        c->audit.op(L.tail, {L.ws_run}, {L.ws_run, L.host_buf}, "MSM pass: checked redo + tail");
        HIPCHK(c, aud_sync(c, L.tail));
    }
    if (L.fixed) {
        // fixed-base mode: one independent result per column
        const uint32_t per = msm_ws_sums_per_result(L.ws_run);
        for (uint32_t q = 0; q < L.batch; q++) out[q] = msm_ws_finish_fixed(L.ws_run, L.host_buf + (size_t)q * per);
    } else {
        out[0] = msm_finish_host(L.host_buf, L.nwin, L.cw);  // generic mode: Horner over the windows
    }
    // timers: head (recode .. accumulate, on the context stream) and the accumulate kernel alone
    // (the head ends with the accumulate kernel: t_acc[1] — always complete once the tail is; an event recorded behind the tail's
    // own on the same stream, as round 4 did, is often not: with the tails on the main stream most passes went uncounted)
    float ms = 0.f;
    c->acc_n[ZK_T_MSM]++;
    if (L.n > 0 && hipEventElapsedTime(&ms, L.t_head[0], L.t_acc[1]) == hipSuccess) {
        c->acc_ms[ZK_T_MSM] += ms;
        c->last_plain_ms[ZK_T_MSM] = ms;
    }
    if (L.n > 0 && hipEventElapsedTime(&ms, L.t_acc[0], L.t_acc[1]) == hipSuccess) {
        c->acc_ms[ZK_T_MSM_ACCUM] += ms;
        c->acc_n[ZK_T_MSM_ACCUM]++;
        c->acc_n[ZK_T_MSM_COLUMNS] += L.batch;
        c->last_plain_ms[ZK_T_MSM_ACCUM] = ms;
    }
    if (L.n > 0 && L.fixed && msm_ws_last_pass_wide(L.ws_run) && hipEventElapsedTime(&ms, L.t_acc[2], L.t_acc[3]) == hipSuccess) {
        c->acc_ms[ZK_T_MSM_TAIL] += ms;
        c->acc_n[ZK_T_MSM_TAIL]++;
        c->last_plain_ms[ZK_T_MSM_TAIL] = ms;
    }
    return ZK_OK;
}

void ctx_msm_drain(zk_ctx* c) {
    for (int q = 0; q < zk_ctx::MSM_LANES; q++)
        if (c->lanes[q].busy) {
            aud_esync(c, c->lanes[q].tail_done);
            c->lanes[q].busy = false;
        }
}

int ctx_msm_end(zk_ctx* c, int lane, G1Jac* out) {
    if (lane >= 0 && lane < zk_ctx::MSM_LANES && c->lanes[lane].busy && c->lanes[lane].batch != 1) return ZK_EINVAL;
    return ctx_msm_end_batch(c, lane, out);
}

// synchronous form (also feeds the accumulated timers used by bench.py)
int ctx_msm_device(zk_ctx* c, const Fr* d_scalars, const G1Affine* d_bases, size_t n, G1Jac* out) {
    int rc = ctx_msm_begin(c, 0, d_scalars, d_bases, n);
    if (rc) return rc;
    return ctx_msm_end(c, 0, out);
}

// ------------------------------------------------------------------ C ABI --

int zk_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// PCI address ("0000:c1:00.0") of a device: where its host-side neighbourhood is found (/sys/bus/pci/devices/<id>/numa_node,
// local_cpulist) — a multi-GPU host binds each GPU's worker threads and staging memory to that NUMA node
ZK_API(zk_device_pci_bus_id, (int device_id, char* out, size_t cap), (device_id, out, cap)) {
    if (!out || cap < 16) return ZK_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ZK_ENODEV;
    if (device_id < 0 || device_id >= ndev) return ZK_EINVAL;
    if (hipDeviceGetPCIBusId(out, (int)cap, device_id) != hipSuccess) return ZK_EHIP;
    return ZK_OK;
}

// free / total memory of a device: what a host sizes its number of resident pipelines by (ecdsa_p256.py)
ZK_API(zk_device_mem_info, (int device_id, size_t* free_bytes, size_t* total_bytes), (device_id, free_bytes, total_bytes)) {
    if (!free_bytes || !total_bytes) return ZK_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ZK_ENODEV;
    if (device_id < 0 || device_id >= ndev) return ZK_EINVAL;
    int prev = 0;
    if (hipGetDevice(&prev) != hipSuccess || hipSetDevice(device_id) != hipSuccess) return ZK_EHIP;
    const hipError_t e = hipMemGetInfo(free_bytes, total_bytes);
    hipSetDevice(prev);
    return e == hipSuccess ? ZK_OK : ZK_EHIP;
}

// page-locked host memory for the buffers a host hands to zk_poly_upload*: the copy is then one DMA at the bus rate instead of
// the runtime's staged copy out of pageable memory (allocate on the thread that is bound to the GPU's NUMA node)
void* zk_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes) != hipSuccess) return nullptr;
    return p;
}
void zk_host_free(void* p) {
    if (p) hipHostFree(p);
}

const char* zk_strerror(int code) {
    switch (code) {
        case ZK_OK: return "ok";
        case ZK_EINVAL: return "invalid argument";
        case ZK_ENOMEM: return "out of memory";
        case ZK_EHIP: return "HIP runtime error";
        case ZK_ENODEV: return "no usable gfx950 device";
        case ZK_ESTATE: return "missing prerequisite (SRS / key not loaded)";
        case ZK_EWITNESS: return "witness does not satisfy the circuit (lookup input outside the table)";
        case ZK_EINTERNAL: return "internal error (C++ exception stopped at the ABI boundary)";
        case ZK_ELAYOUT: return "selector columns outside the layout of compress_selectors the key is built for";
        default: return "unknown error";
    }
}

ZK_API(zk_ctx_create, (int device_id, zk_ctx** out), (device_id, out)) {
    if (!out) return ZK_EINVAL;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ZK_ENODEV;
    if (device_id < 0 || device_id >= ndev) return ZK_EINVAL;
    zk_ctx* c = new (std::nothrow) zk_ctx();
    if (!c) return ZK_ENOMEM;
    c->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess ||
        ((c->stream_slot = pool_take_slot(device_id, &c->stream)) < 0 && hipStreamCreate(&c->stream) != hipSuccess) ||
        hipHostMalloc(&c->host_small, 8 * sizeof(Fr)) != hipSuccess ||
        hipMalloc(&c->small, (2048 + 8) * sizeof(Fr)) != hipSuccess) {
        zk_ctx_destroy(c);
        return ZK_EHIP;
    }
    for (int i = 0; i < ZK_T_COUNT; i++)
        if (hipEventCreate(&c->ev[i][0]) != hipSuccess || hipEventCreate(&c->ev[i][1]) != hipSuccess) {
            zk_ctx_destroy(c);
            return ZK_EHIP;
        }
    // (the transform and MSM streams of a lone proof are made on first use — ctx_lone_streams: the HIP runtime spreads a
    // process's streams over its four hardware queues as they are created, and with four streams per context the MAIN streams of
    // four pipelines all landed on one queue: 100 -> 88 proofs/s, accumulate launches serialised at 0.70 ms, found by bench.py)
    if (hipEventCreateWithFlags(&c->ev_msm_in, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_rows, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_xform, hipEventDisableTiming) != hipSuccess) {
        zk_ctx_destroy(c);
        return ZK_EHIP;
    }
    for (int i = 0; i < zk_ctx::MSM_LANES; i++) {
        zk_ctx::MsmLane& L = c->lanes[i];
        L.tail = c->stream;
        if (hipEventCreate(&L.t_head[0]) != hipSuccess ||
            hipEventCreate(&L.t_head[1]) != hipSuccess || hipEventCreate(&L.t_acc[0]) != hipSuccess ||
            hipEventCreate(&L.t_acc[1]) != hipSuccess || hipEventCreate(&L.t_acc[2]) != hipSuccess || hipEventCreate(&L.t_acc[3]) != hipSuccess || hipEventCreateWithFlags(&L.head_done, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&L.tail_done, hipEventDisableTiming) != hipSuccess ||
            hipHostMalloc(&L.host_buf, (size_t)MSM_MAX_BATCH * 15 * 4 * sizeof(G1X)) != hipSuccess) {
            zk_ctx_destroy(c);
            return ZK_EHIP;
        }
    }
    c->zeta = fr_zeta();
    c->zeta2 = fe_sqr(c->zeta);
    ctx_activity_register(c);
    *out = c;
    return ZK_OK;
}

// A second context on the same device that shares `parent`'s resident SRS (bases + window tables, read-only): the way to run
// several proof pipelines per GPU (one zk_ctx per host thread) without a copy of the tables each.  The child sees the SRS as it
// is NOW; either context may later load another SRS for itself (the shared block lives until its last user lets go).
ZK_API(zk_ctx_create_shared, (zk_ctx* parent, zk_ctx** out), (parent, out)) {
    if (!parent || !out) return ZK_EINVAL;
    zk_ctx* c = nullptr;
    int rc = zk_ctx_create(parent->device, &c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(parent->mu);
    if (parent->srs_k >= 0) {
        c->srs = parent->srs;
        c->srs_k = parent->srs_k;
        c->g = parent->g;
        c->g_lagrange = parent->g_lagrange;
        c->g_table = parent->g_table;
        c->g_lagrange_table = parent->g_lagrange_table;
        c->table_c = parent->table_c;
        c->g_has_identity = parent->g_has_identity;
        c->g_lagrange_has_identity = parent->g_lagrange_has_identity;
        c->g2_valid = parent->g2_valid;
        memcpy(c->g2_raw, parent->g2_raw, 128);
        memcpy(c->s_g2_raw, parent->s_g2_raw, 128);
        c->opt_msm_window = parent->opt_msm_window;  // the tables were built for this window
        c->opt_msm_batch = parent->opt_msm_batch;
        c->opt_ntt_max_r = parent->opt_ntt_max_r;
        c->opt_gp_batch_invert = parent->opt_gp_batch_invert;
        c->opt_tail_stream = parent->opt_tail_stream;
        c->opt_tail_main_above = parent->opt_tail_main_above;
        c->opt_batch_pass_cols = parent->opt_batch_pass_cols;
        c->opt_xform_stream = parent->opt_xform_stream;
        c->opt_msm_stream = parent->opt_msm_stream;
        c->opt_msm_t1 = parent->opt_msm_t1;
        c->srs_gen++;
    }
    *out = c;
    return ZK_OK;
}

void zk_ctx_destroy(zk_ctx* c) {
    if (!c) return;
    ctx_activity_unregister(c);
    hipSetDevice(c->device);
    if (c->stream) aud_sync(c, c->stream);
    if (c->tail_stream) aud_sync(c, c->tail_stream);
    if (c->xform_stream) aud_sync(c, c->xform_stream);
    if (c->msm_stream) aud_sync(c, c->msm_stream);
    for (auto& kv : c->twiddles) hipFree(kv.second);
    for (auto& kv : c->twiddles_ntt) hipFree(kv.second);
    for (auto& kv : c->twiddles_ninv) hipFree(kv.second);
    for (auto& kv : c->coset_points) hipFree(kv.second);
    for (auto& kv : c->coset3_pre) hipFree(kv.second);
    pk_destroy_all(c);
    for (auto& kv : c->polys) hipFree(kv.second.ptr);
    for (auto& r : c->poly_spare) hipFree(r.ptr);
    c->srs.reset();  // frees the bases and tables unless another context shares them
    for (int i = 0; i < zk_ctx::MSM_LANES; i++) {
        zk_ctx::MsmLane& L = c->lanes[i];
        if (L.ws) msm_workspace_destroy(L.ws);
        if (L.ws_gen) msm_workspace_destroy(L.ws_gen);
        if (L.host_buf) hipHostFree(L.host_buf);
        for (int j = 0; j < 2; j++)
            if (L.t_head[j]) hipEventDestroy(L.t_head[j]);
        for (int j = 0; j < 4; j++)
            if (L.t_acc[j]) hipEventDestroy(L.t_acc[j]);
        if (L.head_done) hipEventDestroy(L.head_done);
        if (L.tail_done) hipEventDestroy(L.tail_done);
    }
    if (c->host_small) hipHostFree(c->host_small);
    if (c->scratch) hipFree(c->scratch);
    for (int i = 0; i < 2; i++)
        if (c->seam_buf[i]) hipFree(c->seam_buf[i]);
    if (c->small) hipFree(c->small);
    for (int i = 0; i < ZK_T_COUNT; i++)
        for (int j = 0; j < 2; j++)
            if (c->ev[i][j]) hipEventDestroy(c->ev[i][j]);
    if (c->ev_msm_in) hipEventDestroy(c->ev_msm_in);
    if (c->ev_rows) hipEventDestroy(c->ev_rows);
    if (c->ev_xform) hipEventDestroy(c->ev_xform);
    if (c->stream_slot >= 0) {
        // the slot's streams stay (engine.hip stream pool); a main stream made at its own priority is the context's own
        if (c->stream_own_priority && c->stream) hipStreamDestroy(c->stream);
        pool_release_slot(c->device, c->stream_slot);
    } else {
        if (c->msm_stream) hipStreamDestroy(c->msm_stream);
        if (c->xform_stream) hipStreamDestroy(c->xform_stream);
        if (c->tail_stream) hipStreamDestroy(c->tail_stream);
        if (c->stream) hipStreamDestroy(c->stream);
    }
    delete c;
}

int zk_last_hip_error(const zk_ctx* c) { return c ? c->last_hip : 0; }

ZK_API(zk_sync, (zk_ctx* c), (c)) {
    if (!c) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    int rc = ctx_bind(c);
    if (rc) return rc;
    HIPCHK(c, aud_sync(c, c->stream));
    return ZK_OK;
}

ZK_API(zk_last_kernel_ms, (zk_ctx* c, int which, float* out_ms), (c, which, out_ms)) {
    if (!c || !out_ms || which < 0 || which >= ZK_T_COUNT) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    if (which == ZK_T_MSM || which == ZK_T_MSM_ACCUM) {
        *out_ms = c->last_plain_ms[which];
        return ZK_OK;
    }
    if (!c->ev_valid[which]) {
        *out_ms = 0.f;
        return ZK_OK;
    }
    int rc = ctx_bind(c);
    if (rc) return rc;
    HIPCHK(c, aud_esync(c, c->ev[which][1]));
    HIPCHK(c, hipEventElapsedTime(out_ms, c->ev[which][0], c->ev[which][1]));
    return ZK_OK;
}

// ---- shader-clock probe (bench.py's roofline.valu_issue): ONE wave spins for `ticks` of the constant 100 MHz counter
// (s_memrealtime) on a chain of dependent v_mad_u64_u32 and reports what the shader-clock counter (s_memtime) advanced by in
// the same interval.  Run on a context of its own WHILE the workload proves, it reads the clock the chip sustains under that
// load (a lone probe on an idle chip reads the boost clock).  out[0] = shader-clock ticks, out[1] = 100 MHz ticks, out[2] =
// multiply-adds of the chain (one wave, nothing to interleave with: ticks / mads = the multiplier's dependent latency)
__global__ __launch_bounds__(64) void clock_probe_kernel(uint64_t ticks, uint64_t* __restrict__ out) {
    uint64_t acc = threadIdx.x + 1;
    uint32_t a = 0x9e3779b9u + threadIdx.x, b = 0x7f4a7c15u;
    const uint64_t r0 = wall_clock64();
    const uint64_t c0 = clock64();
    uint64_t iters = 0;
    while (wall_clock64() - r0 < ticks) {
#pragma unroll
        for (int i = 0; i < 256; i++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
        iters += 256;
    }
    const uint64_t c1 = clock64();
    const uint64_t r1 = wall_clock64();
    if (threadIdx.x == 0) {
        out[0] = c1 - c0;
        out[1] = r1 - r0;
        out[2] = iters;
        out[3] = acc;  // (keeps the chain alive)
    }
}

ZK_API(zk_clock_probe, (zk_ctx* c, uint32_t millis, uint64_t out[4]), (c, millis, out)) {
    if (!c || !out || millis == 0 || millis > 2000) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    int rc = ctx_bind(c);
    if (rc) return rc;
    uint64_t* d = reinterpret_cast<uint64_t*>(c->small);         // device scratch of the context (8 field elements)
    uint64_t* h = reinterpret_cast<uint64_t*>(c->host_small);    // its pinned twin
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, c->stream, (uint64_t)millis * 100000ull, d);
    HIPCHK(c, hipMemcpyAsync(h, d, 4 * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, aud_sync(c, c->stream));
    memcpy(out, h, 4 * sizeof(uint64_t));
    return ZK_OK;
}

ZK_API(zk_audit_report, (zk_ctx* c, uint64_t counts[2], char* msg, size_t cap), (c, counts, msg, cap)) {
    if (!c || !counts) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    counts[0] = c->audit.checks;
    counts[1] = c->audit.violations;
    if (msg && cap) {
        const size_t len = std::min(cap - 1, c->audit.first.size());
        memcpy(msg, c->audit.first.data(), len);
        msg[len] = 0;
    }
    return ZK_OK;
}

ZK_API(zk_timer_reset, (zk_ctx* c), (c)) {
    if (!c) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    for (int i = 0; i < ZK_T_COUNT; i++) {
        c->acc_ms[i] = 0;
        c->acc_n[i] = 0;
    }
    return ZK_OK;
}

ZK_API(zk_timer_stats, (zk_ctx* c, int which, double* total_ms, uint64_t* count), (c, which, total_ms, count)) {
    if (!c || which < 0 || which >= ZK_T_COUNT || !total_ms || !count) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    *total_ms = c->acc_ms[which];
    *count = c->acc_n[which];
    return ZK_OK;
}

ZK_API(zk_ctx_set_option, (zk_ctx* c, int option, int64_t value), (c, option, value)) {
    if (!c || value < 0) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    switch (option) {
        case ZK_OPT_MSM_WINDOW:
            if (value && (value < 9 || value > 17)) return ZK_EINVAL;
            c->opt_msm_window = (uint32_t)value;
            return ZK_OK;
        case ZK_OPT_MSM_TAIL_STREAM:
            if (value > 2) return ZK_EINVAL;
            c->opt_tail_stream = (uint32_t)value;
            return ZK_OK;
        case ZK_OPT_MSM_TAIL_MAIN_ABOVE:
            if (value > 64) return ZK_EINVAL;
            c->opt_tail_main_above = (uint32_t)value;
            return ZK_OK;
        case ZK_OPT_MSM_STREAM:
            if (value > 2) return ZK_EINVAL;
            c->opt_msm_stream = (uint32_t)value;
            return ZK_OK;
        case ZK_OPT_MSM_T1:
            if (value > 2) return ZK_EINVAL;
            c->opt_msm_t1 = (uint32_t)value;
            return ZK_OK;
        case ZK_OPT_STREAM_AUDIT:
            if (value > 2) return ZK_EINVAL;
            c->audit.reset();
            c->audit.streams[0] = c->stream;
            c->audit.on = value != 0;
            c->audit_fault = value == 2;
            return ZK_OK;
        case ZK_OPT_XFORM_STREAM:
            if (value > 2) return ZK_EINVAL;
            c->opt_xform_stream = (uint32_t)value;
            return ZK_OK;
        case ZK_OPT_BATCH_PASS_COLUMNS:
            if (value > MSM_MAX_BATCH) return ZK_EINVAL;
            c->opt_batch_pass_cols = (uint32_t)value;
            return ZK_OK;
        case ZK_OPT_MSM_BATCH:
            if (value > MSM_MAX_BATCH) return ZK_EINVAL;
            c->opt_msm_batch = (uint32_t)value;
            return ZK_OK;
        case ZK_OPT_NTT_MAX_RADIX_LOG2:
            if (value && (value < 1 || value > 11)) return ZK_EINVAL;  // clamped to the tile size in ntt_run
            c->opt_ntt_max_r = (uint32_t)value;
            return ZK_OK;
        case ZK_OPT_GP_BATCH_INVERT:
            c->opt_gp_batch_invert = value ? 1 : 0;
            return ZK_OK;
        case ZK_OPT_ACTIVITY_HOLD:
            if (value > 2) return ZK_EINVAL;
            c->opt_no_activity_hold = value == 1;
            // 2: held from now on, whatever the entry points used (a phase-level host's worker context); 0 / 1 let go of such a hold
            c->opt_activity_pinned = value == 2;
            if (!c->act_held || value != 2) ctx_activity_hold(c, false);
            if (value == 2) ctx_activity_hold(c, true);
            return ZK_OK;
        case ZK_OPT_QUOTIENT_DOMAIN:
            if (value > 2) return ZK_EINVAL;
            c->opt_quotient_domain = (uint32_t)value;
            return ZK_OK;
        case ZK_OPT_STREAM_PRIORITY: {
            // experiment (docs/experiments.md "pipelines at different priorities"): the context's MAIN stream is made again at
            // another dispatch priority.  Only meaningful before any work was enqueued (the audit ledger and the lone-proof side
            // streams refer to the main stream by value); the old stream is drained first
            if (value > 2) return ZK_EINVAL;
            for (int i = 0; i < zk_ctx::MSM_LANES; i++)
                if (c->lanes[i].busy) return ZK_EINVAL;  // an MSM pass in flight holds the stream by value
            int rc = ctx_bind(c);
            if (rc) return rc;
            int lo = 0, hi = 0;  // numerically lower = higher priority
            if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) return ZK_EHIP;
            const int pr = value == 1 ? hi : value == 2 ? lo : (lo + hi) / 2;
            hipStream_t ns = nullptr;
            if (hipStreamCreateWithPriority(&ns, hipStreamDefault, pr) != hipSuccess) return ZK_EHIP;
            hipStreamSynchronize(c->stream);
            for (int i = 0; i < zk_ctx::MSM_LANES; i++)
                if (c->lanes[i].tail == c->stream) c->lanes[i].tail = ns;
            if (c->stream_own_priority || c->stream_slot < 0) hipStreamDestroy(c->stream);  // (a slot's main stream stays in its slot)
            c->stream = ns;
            c->stream_own_priority = true;
            c->audit.streams[0] = ns;
            return ZK_OK;
        }
        default: return ZK_EINVAL;
    }
}

// ---- fine-grained seam -------------------------------------------------------

ZK_API(zk_msm_bn254, (zk_ctx* c, const uint64_t* scalars, const uint64_t* bases, size_t n, uint64_t out[12]), (c, scalars, bases, n, out)) {
    if (!c || !out || (n && (!scalars || !bases))) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    int rc = ctx_bind(c);
    if (rc) return rc;
    G1Jac res;
    if (n == 0) {
        res.x = Fq::one();
        res.y = Fq::one();
        res.z = Fq::zero();
        memcpy(out, &res, 96);
        return ZK_OK;
    }
    Fr* d_s = nullptr;
    G1Affine* d_b = nullptr;
    if ((rc = seam_buffer(c, 0, n * sizeof(Fr), (void**)&d_s))) return rc;
    // arbitrary bases: both operands are uploaded on every call (the resident-SRS form is zk_msm_srs)
    if (hipMemcpyAsync(d_s, scalars, n * sizeof(Fr), hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = ZK_EHIP;
    if (rc == ZK_OK) {
        if ((rc = seam_buffer(c, 1, n * sizeof(G1Affine), (void**)&d_b))) return rc;
        if (hipMemcpyAsync(d_b, bases, n * sizeof(G1Affine), hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = ZK_EHIP;
    }
    if (rc == ZK_OK) rc = ctx_msm_device(c, d_s, d_b, n, &res);
    aud_sync(c, c->stream);
    if (rc == ZK_OK) memcpy(out, &res, 96);
    return rc;
}

ZK_API(zk_msm_srs, (zk_ctx* c, int basis, const uint64_t* scalars, size_t n, uint64_t out[12]), (c, basis, scalars, n, out)) {
    if (!c || !out || (n && !scalars) || (basis != ZK_BASIS_MONOMIAL && basis != ZK_BASIS_LAGRANGE)) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->srs_k < 0) return ZK_ESTATE;
    if (n > ((size_t)1 << c->srs_k)) return ZK_EINVAL;
    int rc = ctx_bind(c);
    if (rc) return rc;
    G1Jac res;
    if (n == 0) {
        res.x = Fq::one();
        res.y = Fq::one();
        res.z = Fq::zero();
        memcpy(out, &res, 96);
        return ZK_OK;
    }
    Fr* d_s = nullptr;
    if ((rc = seam_buffer(c, 0, n * sizeof(Fr), (void**)&d_s))) return rc;
    if (hipMemcpyAsync(d_s, scalars, n * sizeof(Fr), hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = ZK_EHIP;
    if (rc == ZK_OK) rc = ctx_msm_device(c, d_s, basis == ZK_BASIS_LAGRANGE ? c->g_lagrange : c->g, n, &res);
    aud_sync(c, c->stream);
    if (rc == ZK_OK) memcpy(out, &res, 96);
    return rc;
}

ZK_API(zk_ntt_bn254_fr, (zk_ctx* c, uint64_t* a, const uint64_t omega[4], uint32_t log_n), (c, a, omega, log_n)) {
    if (!c || !a || !omega || log_n > 26) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    int rc = ctx_bind(c);
    if (rc) return rc;
    const size_t n = (size_t)1 << log_n;
    Fr w;
    memcpy(&w, omega, 32);
    // the standard root or its inverse use the cached table; anything else gets a one-off table
    const Fr std_w = fr_omega(log_n);
    const Fr* tw = nullptr;
    Fr* own_tw = nullptr;
    uint32_t inverse = 0;
    if (w == std_w) {
        rc = ctx_get_twiddles_ntt(c, log_n, &tw);
    } else if (fe_mul(w, std_w) == Fr::one()) {
        rc = ctx_get_twiddles_ntt(c, log_n, &tw);
        inverse = 1;
    } else {
        if (hipMalloc(&own_tw, n * sizeof(Fr)) != hipSuccess) return ZK_ENOMEM;
        launch_twiddles_internal(own_tw, w, (uint32_t)n, c->stream);
        tw = own_tw;
    }
    if (rc) return rc;
    Fr *d_a = nullptr, *d_t = nullptr;
    if ((rc = seam_buffer(c, 0, n * sizeof(Fr), (void**)&d_a)) || (rc = seam_buffer(c, 1, n * sizeof(Fr), (void**)&d_t))) {
        hipFree(own_tw);
        return rc;
    }
    NttJob job;
    memset(&job, 0, sizeof(job));
    job.src = d_a;
    job.dst = d_a;
    job.tmp = d_t;
    job.tw = tw;
    job.log_n = log_n;
    job.inverse = inverse;
    job.n_in = job.n_out = (uint32_t)n;
    job.max_log_r = c->opt_ntt_max_r;
    if (!own_tw && log_n > 7) rc = ctx_get_twiddles(c, log_n, &job.tw_last);  // best_fft does not scale: c = 1 in both directions
    if (rc == ZK_OK && hipMemcpyAsync(d_a, a, n * sizeof(Fr), hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = ZK_EHIP;
    if (rc == ZK_OK) {
        aud_record(c, c->ev[ZK_T_NTT][0], c->stream);
        hipError_t e = ntt_run(job, c->stream);
        aud_record(c, c->ev[ZK_T_NTT][1], c->stream);
        c->ev_valid[ZK_T_NTT] = true;
        if (e != hipSuccess) {
            c->last_hip = (int)e;
            rc = ZK_EHIP;
        }
    }
    // The transform is complete (and known to have succeeded) BEFORE the first byte of the caller's buffer is overwritten: a
    // failed upload or kernel leaves `a` untouched.  (Round 4 staged the result through a zero-filled temporary — 64 MiB of page
    // faults and a second copy per 2^21 call: 19 ms of the call's 19.4; what can still fail below is the copy-out itself.)
    if (rc == ZK_OK && (aud_sync(c, c->stream) != hipSuccess || hipGetLastError() != hipSuccess)) rc = ZK_EHIP;
    if (rc == ZK_OK && (hipMemcpyAsync(a, d_a, n * sizeof(Fr), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
                        aud_sync(c, c->stream) != hipSuccess))
        rc = ZK_EHIP;
    aud_sync(c, c->stream);
    hipFree(own_tw);
    return rc;
}

// ---- SRS ---------------------------------------------------------------------
// window-multiple tables of both bases for the fixed-base MSM (k >= 10; smaller SRS use the generic path)
int srs_build_tables(zk_ctx* c, uint32_t k) {
    if (k < 10) return ZK_OK;
    const uint32_t n = 1u << k;
    const uint32_t cw = msm_auto_window(n, c->opt_msm_window);
    const size_t cnt = (size_t)msm_num_windows(cw) * n;
    if (hipMalloc(&c->srs->g_table, cnt * sizeof(G1Affine)) != hipSuccess ||
        hipMalloc(&c->srs->g_lagrange_table, cnt * sizeof(G1Affine)) != hipSuccess)
        return ZK_ENOMEM;
    c->g_table = c->srs->g_table;
    c->g_lagrange_table = c->srs->g_lagrange_table;
    hipError_t e = msm_build_table(c->g, n, cw, c->g_table, c->stream);
    if (e == hipSuccess) e = msm_build_table(c->g_lagrange, n, cw, c->g_lagrange_table, c->stream);
    if (e == hipSuccess) e = msm_bases_have_identity(c->g, n, c->stream, (uint32_t*)c->small, (uint32_t*)c->host_small, &c->g_has_identity);
    if (e == hipSuccess)
        e = msm_bases_have_identity(c->g_lagrange, n, c->stream, (uint32_t*)c->small, (uint32_t*)c->host_small, &c->g_lagrange_has_identity);
    if (e == hipSuccess) e = aud_sync(c, c->stream);
    if (e != hipSuccess) {
        c->last_hip = (int)e;
        return ZK_EHIP;
    }
    c->table_c = cw;
    return ZK_OK;
}

// a fresh, empty block for this context (the previous one is released: freed unless another context still shares it)
static void srs_new_block(zk_ctx* c) {
    c->g2_valid = false;
    c->srs_gen++;  // proving keys made under the previous SRS are refused from now on (ZK_ESTATE)
    c->srs = std::make_shared<SrsBlock>();
    c->srs->device = c->device;
    c->g = c->g_lagrange = c->g_table = c->g_lagrange_table = nullptr;
    c->table_c = 0;
    c->srs_k = -1;
}

void srs_adopt(zk_ctx* c, uint32_t k, G1Affine* g, G1Affine* g_lagrange) {
    (void)k;
    srs_new_block(c);
    c->g = c->srs->g = g;
    c->g_lagrange = c->srs->g_lagrange = g_lagrange;
}

int srs_alloc(zk_ctx* c, uint32_t k) {
    if (k < 1 || k > 24) return ZK_EINVAL;
    const size_t n = (size_t)1 << k;
    srs_new_block(c);
    if (hipMalloc(&c->srs->g, n * sizeof(G1Affine)) != hipSuccess || hipMalloc(&c->srs->g_lagrange, n * sizeof(G1Affine)) != hipSuccess)
        return ZK_ENOMEM;
    c->g = c->srs->g;
    c->g_lagrange = c->srs->g_lagrange;
    return ZK_OK;
}

ZK_API(zk_srs_setup, (zk_ctx* c, uint32_t k, const uint8_t seed[32]), (c, k, seed)) {
    if (!c || !seed) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    int rc = ctx_bind(c);
    if (rc) return rc;
    ctx_release_spares(c);  // parked vectors are reclaimable: give them back before the big allocations
    if ((rc = srs_alloc(c, k)) != ZK_OK) return rc;
    const uint32_t n = 1u << k;
    ChaCha20Rng rng(seed);
    const Fr s = rng.next_fr();
    // window-8 table of the generator on the host: table[w*256 + d] = [d * 256^w] G1
    std::vector<G1Affine> table(32 * 256);
    {
        std::vector<G1X> acc(32 * 256);
        G1X base;
        base.x = Fq::one();
        base.y = fe_add(Fq::one(), Fq::one());
        base.zz = Fq::one();
        base.zzz = Fq::one();
        for (int w = 0; w < 32; w++) {
            G1X cur = G1X::identity();
            acc[w * 256] = cur;
            for (int d = 1; d < 256; d++) {
                g1x_add(cur, base);
                acc[w * 256 + d] = cur;
            }
            g1x_add(cur, base);
            base = cur;
        }
        // batch normalisation: invert all ZZZ at once
        std::vector<Fq> pref(acc.size() + 1);
        pref[0] = Fq::one();
        for (size_t i = 0; i < acc.size(); i++) pref[i + 1] = acc[i].is_identity() ? pref[i] : fe_mul(pref[i], acc[i].zzz);
        Fq inv = fe_inv(pref[acc.size()]);
        for (size_t i = acc.size(); i-- > 0;) {
            if (acc[i].is_identity()) {
                table[i].x = Fq::zero();
                table[i].y = Fq::zero();
                continue;
            }
            const Fq t = fe_mul(inv, pref[i]);
            inv = fe_mul(inv, acc[i].zzz);
            const Fq u = fe_mul(acc[i].zz, t);
            table[i].x = fe_mul(acc[i].x, fe_sqr(u));
            table[i].y = fe_mul(acc[i].y, t);
        }
    }
    G1Affine* d_table = nullptr;
    Fr* d_sc = nullptr;
    if (hipMalloc(&d_table, table.size() * sizeof(G1Affine)) != hipSuccess || hipMalloc(&d_sc, (size_t)n * sizeof(Fr)) != hipSuccess) {
        hipFree(d_table);
        return ZK_ENOMEM;
    }
    rc = ZK_OK;
    const Fr* tw = nullptr;
    if (hipMemcpyAsync(d_table, table.data(), table.size() * sizeof(G1Affine), hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = ZK_EHIP;
    if (rc == ZK_OK) rc = ctx_get_twiddles(c, k, &tw);
    if (rc == ZK_OK) {
        // g[i] = [s^i] G
        launch_twiddles(d_sc, s, n, c->stream);
        launch_srs_fixed_base(d_sc, n, d_table, c->g, c->stream);
        // g_lagrange[i] = [L_i(s)] G,  L_i(s) = w^i (s^n - 1) / (n (s - w^i))
        Fr sn = s;
        for (uint32_t i = 0; i < k; i++) sn = fe_sqr(sn);
        const Fr cst = fe_mul(fe_sub(sn, Fr::one()), fe_inv(fr_from_u64(n)));
        launch_srs_lagrange_scalars(tw, n, s, cst, d_sc, c->stream);
        launch_srs_fixed_base(d_sc, n, d_table, c->g_lagrange, c->stream);
        hipError_t e = aud_sync(c, c->stream);
        if (e == hipSuccess) e = hipGetLastError();
        if (e != hipSuccess) {
            c->last_hip = (int)e;
            rc = ZK_EHIP;
        }
    }
    hipFree(d_table);
    hipFree(d_sc);
    if (rc == ZK_OK) rc = srs_build_tables(c, k);
    if (rc == ZK_OK) {
        srs_set_g2_from_secret(c, s);  // g2 = G2 generator, s_g2 = [s]G2 (host side: it only travels through zk_srs_write)
        c->srs_k = (int)k;
    }
    return rc;
}

ZK_API(zk_srs_load, (zk_ctx* c, uint32_t k, const uint64_t* g, const uint64_t* gl), (c, k, g, gl)) {
    if (!c || !g || !gl) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    int rc = ctx_bind(c);
    if (rc) return rc;
    ctx_release_spares(c);
    if ((rc = srs_alloc(c, k)) != ZK_OK) return rc;
    const size_t bytes = ((size_t)1 << k) * sizeof(G1Affine);
    HIPCHK(c, hipMemcpy(c->g, g, bytes, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->g_lagrange, gl, bytes, hipMemcpyHostToDevice));
    if ((rc = srs_build_tables(c, k)) != ZK_OK) return rc;
    c->srs_k = (int)k;
    return ZK_OK;
}

ZK_API(zk_srs_export, (zk_ctx* c, int basis, uint64_t* out, size_t first, size_t count), (c, basis, out, first, count)) {
    if (!c || !out) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->srs_k < 0) return ZK_ESTATE;
    const size_t n = (size_t)1 << c->srs_k;
    if (first > n || count > n - first) return ZK_EINVAL;
    int rc = ctx_bind(c);
    if (rc) return rc;
    const G1Affine* src = basis == ZK_BASIS_LAGRANGE ? c->g_lagrange : c->g;
    HIPCHK(c, hipMemcpy(out, src + first, count * sizeof(G1Affine), hipMemcpyDeviceToHost));
    return ZK_OK;
}

int zk_srs_k(const zk_ctx* c) { return c ? c->srs_k : -1; }

ZK_API(zk_srs_msm_plan, (const zk_ctx* c, uint32_t* window_bits, uint32_t* windows), (c, window_bits, windows)) {
    if (!c || !window_bits || !windows) return ZK_EINVAL;
    if (c->srs_k < 0) return ZK_ESTATE;
    *window_bits = c->table_c;  // 0: no window-multiple tables (k < 10), zk_commit takes the generic path
    *windows = c->table_c ? msm_num_windows(c->table_c) : 0;
    return ZK_OK;
}

// ---- resident polynomials ----------------------------------------------------

// Vectors parked by zk_poly_free (up to POLY_SPARE_BYTES per context) are reclaimable memory: the large allocators
// (zk_keygen, zk_srs_setup / load / read, zk_pk_read, the MSM workspaces) release them up front instead of failing with
// ZK_ENOMEM while they sit idle.  Caller holds c->mu and has bound the device.
void ctx_release_spares(zk_ctx* c) {
    for (auto& r : c->poly_spare) hipFree(r.ptr);
    c->poly_spare.clear();
    c->poly_spare_bytes = 0;
}

static PolyRec* find_poly(zk_ctx* c, zk_poly h) {
    auto it = c->polys.find(h);
    return it == c->polys.end() ? nullptr : &it->second;
}

ZK_API(zk_poly_alloc, (zk_ctx* c, size_t n, zk_poly* out), (c, n, out)) {
    if (!c || !out || n == 0) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    int rc = ctx_bind(c);
    if (rc) return rc;
    Fr* p = nullptr;
    for (size_t i = 0; i < c->poly_spare.size(); i++)
        if (c->poly_spare[i].n == n) {  // a vector of this length given back earlier (contents undefined, as hipMalloc's)
            p = c->poly_spare[i].ptr;
            c->poly_spare_bytes -= n * sizeof(Fr);
            c->poly_spare.erase(c->poly_spare.begin() + i);
            break;
        }
    if (!p && hipMalloc(&p, n * sizeof(Fr)) != hipSuccess) {
        // out of memory with vectors parked: let them go and try once more
        ctx_release_spares(c);
        (void)hipGetLastError();
        if (hipMalloc(&p, n * sizeof(Fr)) != hipSuccess) return ZK_ENOMEM;
    }
    const uint64_t h = c->next_handle++;
    c->polys[h] = PolyRec{p, n};
    *out = h;
    return ZK_OK;
}

ZK_API(zk_poly_free, (zk_ctx* c, zk_poly h), (c, h)) {
    if (!c) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    PolyRec* r = find_poly(c, h);
    if (!r) return ZK_EINVAL;
    ctx_bind(c);
    aud_sync(c, c->stream);  // nothing of this context still uses it
    const size_t bytes = r->n * sizeof(Fr);
    if (c->poly_spare.size() < zk_ctx::POLY_SPARE_MAX && c->poly_spare_bytes + bytes <= zk_ctx::POLY_SPARE_BYTES) {
        c->poly_spare.push_back(*r);
        c->poly_spare_bytes += bytes;
    } else {
        hipFree(r->ptr);
    }
    c->polys.erase(h);
    return ZK_OK;
}

// Handing a resident vector from one context to another of the same device, without a copy and without either context
// waiting on the other's lock: the owner detaches it (the record leaves the context for a process-wide table), the new owner
// attaches it.  A loader context (its own stream and host thread, e.g. one made with zk_ctx_create_shared) can thus upload
// and convert the next request's advice columns while the proving context is inside zk_prove, and the prover's thread picks
// them up between two proofs.
namespace {
struct Detached {
    PolyRec rec;
    int device;
};
std::mutex g_detached_mu;
std::map<uint64_t, Detached> g_detached;
uint64_t g_detached_next = 1;
}  // namespace

ZK_API(zk_poly_detach, (zk_ctx* c, zk_poly h, uint64_t* token), (c, h, token)) {
    if (!c || !token) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    PolyRec* r = find_poly(c, h);
    if (!r) return ZK_EINVAL;
    int rc = ctx_bind(c);
    if (rc) return rc;
    HIPCHK(c, aud_sync(c, c->stream));  // whatever this context still does with the vector finishes first
    const Detached d{*r, c->device};
    c->polys.erase(h);
    std::lock_guard<std::mutex> lg(g_detached_mu);
    *token = g_detached_next++;
    g_detached[*token] = d;
    return ZK_OK;
}

ZK_API(zk_poly_attach, (zk_ctx* c, uint64_t token, zk_poly* out), (c, token, out)) {
    if (!c || !out) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    Detached d;
    {
        std::lock_guard<std::mutex> lg(g_detached_mu);
        auto it = g_detached.find(token);
        if (it == g_detached.end() || it->second.device != c->device) return ZK_EINVAL;
        d = it->second;
        g_detached.erase(it);
    }
    const uint64_t nh = c->next_handle++;
    c->polys[nh] = d.rec;
    *out = nh;
    return ZK_OK;
}

// A detached vector that will never be attached (the loader failed between stage and adopt, the target context is gone):
// without this the record — and its device memory — would stay in the process-wide table for the life of the process.
ZK_API(zk_poly_discard, (uint64_t token), (token)) {
    Detached d;
    {
        std::lock_guard<std::mutex> lg(g_detached_mu);
        auto it = g_detached.find(token);
        if (it == g_detached.end()) return ZK_EINVAL;
        d = it->second;
        g_detached.erase(it);
    }
    int prev = -1;
    hipGetDevice(&prev);
    if (hipSetDevice(d.device) != hipSuccess) return ZK_EHIP;
    const hipError_t e = hipFree(d.rec.ptr);  // hipFree waits for the device: nothing still uses the vector
    if (prev >= 0) hipSetDevice(prev);
    return e == hipSuccess ? ZK_OK : ZK_EHIP;
}

ZK_API(zk_poly_len, (zk_ctx* c, zk_poly h, size_t* out), (c, h, out)) {
    if (!c || !out) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    PolyRec* r = find_poly(c, h);
    if (!r) return ZK_EINVAL;
    *out = r->n;
    return ZK_OK;
}

ZK_API(zk_poly_upload, (zk_ctx* c, zk_poly h, const uint64_t* host, size_t n), (c, h, host, n)) {
    if (!c || !host) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    PolyRec* r = find_poly(c, h);
    if (!r || n > r->n) return ZK_EINVAL;
    int rc = ctx_bind(c);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(r->ptr, host, n * sizeof(Fr), hipMemcpyHostToDevice, c->stream));
    if (n < r->n) HIPCHK(c, hipMemsetAsync(r->ptr + n, 0, (r->n - n) * sizeof(Fr), c->stream));
    HIPCHK(c, aud_sync(c, c->stream));
    return ZK_OK;
}

ZK_API(zk_poly_download, (zk_ctx* c, zk_poly h, uint64_t* host, size_t n), (c, h, host, n)) {
    if (!c || !host) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    PolyRec* r = find_poly(c, h);
    if (!r || n > r->n) return ZK_EINVAL;
    int rc = ctx_bind(c);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(host, r->ptr, n * sizeof(Fr), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, aud_sync(c, c->stream));
    return ZK_OK;
}

// rows [first, first + count) of a resident vector from the host (Montgomery images): the blinding rows a host appends to a
// column the device made (a', s', z), without shipping the column
ZK_API(zk_poly_upload_range, (zk_ctx* c, zk_poly h, size_t first, const uint64_t* host, size_t count), (c, h, first, host, count)) {
    if (!c || (!host && count)) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    PolyRec* r = find_poly(c, h);
    if (!r || first > r->n || count > r->n - first) return ZK_EINVAL;
    int rc = ctx_bind(c);
    if (rc) return rc;
    if (count == 0) return ZK_OK;
    HIPCHK(c, hipMemcpyAsync(r->ptr + first, host, count * sizeof(Fr), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, aud_sync(c, c->stream));
    return ZK_OK;
}

// dst[dst_first ..] = src[src_first .. src_first + count): e.g. the h pieces, n-coefficient slices of the quotient
ZK_API(zk_poly_copy_range, (zk_ctx* c, zk_poly dst, size_t dst_first, zk_poly src, size_t src_first, size_t count), (c, dst, dst_first, src, src_first, count)) {
    if (!c) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    PolyRec *d = find_poly(c, dst), *s = find_poly(c, src);
    if (!d || !s || dst_first > d->n || count > d->n - dst_first || src_first > s->n || count > s->n - src_first) return ZK_EINVAL;
    if (d == s && !(dst_first + count <= src_first || src_first + count <= dst_first)) return ZK_EINVAL;  // overlapping ranges
    int rc = ctx_bind(c);
    if (rc) return rc;
    if (count == 0) return ZK_OK;
    HIPCHK(c, hipMemcpyAsync(d->ptr + dst_first, s->ptr + src_first, count * sizeof(Fr), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, aud_sync(c, c->stream));
    return ZK_OK;
}

ZK_API(zk_poly_copy, (zk_ctx* c, zk_poly dst, zk_poly src), (c, dst, src)) {
    if (!c) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    PolyRec *d = find_poly(c, dst), *s = find_poly(c, src);
    if (!d || !s || d->n < s->n) return ZK_EINVAL;
    int rc = ctx_bind(c);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(d->ptr, s->ptr, s->n * sizeof(Fr), hipMemcpyDeviceToDevice, c->stream));
    if (d->n > s->n) HIPCHK(c, hipMemsetAsync(d->ptr + s->n, 0, (d->n - s->n) * sizeof(Fr), c->stream));
    return ZK_OK;
}

ZK_API(zk_commit, (zk_ctx* c, zk_poly h, int basis, uint64_t out[8]), (c, h, basis, out)) {
    if (!c || !out || (basis != ZK_BASIS_MONOMIAL && basis != ZK_BASIS_LAGRANGE)) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->srs_k < 0) return ZK_ESTATE;
    PolyRec* r = find_poly(c, h);
    const size_t n = (size_t)1 << c->srs_k;
    if (!r || r->n > n) return ZK_EINVAL;
    int rc = ctx_bind(c);
    if (rc) return rc;
    G1Jac j;
    rc = ctx_msm_device(c, r->ptr, basis == ZK_BASIS_LAGRANGE ? c->g_lagrange : c->g, r->n, &j);
    if (rc) return rc;
    const G1Affine a = g1_jac_to_affine_host(j);
    memcpy(out, &a, 64);
    return ZK_OK;
}

ZK_API(zk_commit_batch, (zk_ctx* c, const zk_poly* hs, size_t count, int basis, uint64_t* out), (c, hs, count, basis, out)) {
    if (!c || !out || !hs || count == 0 || (basis != ZK_BASIS_MONOMIAL && basis != ZK_BASIS_LAGRANGE)) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->srs_k < 0) return ZK_ESTATE;
    const size_t n = (size_t)1 << c->srs_k;
    std::vector<const Fr*> ptrs(count);
    size_t len = 0;
    for (size_t i = 0; i < count; i++) {
        PolyRec* r = find_poly(c, hs[i]);
        if (!r || r->n > n || (i && r->n != len)) return ZK_EINVAL;  // one length per call
        len = r->n;
        ptrs[i] = r->ptr;
    }
    int rc = ctx_bind(c);
    if (rc) return rc;
    const G1Affine* bases = basis == ZK_BASIS_LAGRANGE ? c->g_lagrange : c->g;
    const uint32_t cap = ctx_msm_max_batch(c);
    G1Jac js[MSM_MAX_BATCH];
    for (size_t i0 = 0; i0 < count; i0 += cap) {
        const uint32_t cnt = (uint32_t)(count - i0 < cap ? count - i0 : cap);
        rc = ctx_msm_begin_batch(c, 0, ptrs.data() + i0, cnt, bases, len);
        if (rc) return rc;
        rc = ctx_msm_end_batch(c, 0, js);
        if (rc) return rc;
        for (uint32_t q = 0; q < cnt; q++) {
            const G1Affine a = g1_jac_to_affine_host(js[q]);
            memcpy(out + (i0 + q) * 8, &a, 64);
        }
    }
    return ZK_OK;
}


int ctx_ntt(zk_ctx* c, const Fr* src, size_t src_n, Fr* dst, uint32_t log_n, bool inverse, bool coset, size_t n_out) {
    return ctx_ntt_batch(c, &src, src_n, &dst, 1, log_n, inverse, coset, n_out);
}

uint32_t ctx_ntt_max_batch(uint32_t log_n) {
    // scratch for the ping-pong is batch x 2^log_n elements: keep it within 2^23 (256 MiB)
    const uint32_t cap = log_n >= 23 ? 1u : 1u << (23 - log_n);
    return cap < NTT_MAX_BATCH ? cap : NTT_MAX_BATCH;
}

int ctx_ntt_batch(zk_ctx* c, const Fr* const* srcs, size_t src_n, Fr* const* dsts, uint32_t batch, uint32_t log_n, bool inverse,
                  bool coset, size_t n_out, hipStream_t on) {
    const hipStream_t st = on ? on : c->stream;
    const size_t N = (size_t)1 << log_n;
    if (batch == 0 || batch > ctx_ntt_max_batch(log_n)) return ZK_EINVAL;
    int rc = ctx_ensure_scratch(c, N * batch);
    if (rc) return rc;
    const Fr* tw;
    if ((rc = ctx_get_twiddles_ntt(c, log_n, &tw)) != ZK_OK) return rc;
    NttJob job;
    memset(&job, 0, sizeof(job));
    job.batch = batch;
    for (uint32_t b = 0; b < batch; b++) {
        job.srcs[b] = srcs[b];
        job.dsts[b] = dsts[b];
    }
    job.tmp = c->scratch;
    job.tw = tw;
    job.log_n = log_n;
    job.inverse = inverse ? 1 : 0;
    job.n_in = (uint32_t)(src_n < N ? src_n : N);
    job.n_out = (uint32_t)n_out;
    job.max_log_r = c->opt_ntt_max_r;
    if (log_n > 7) {
        // two or more passes: the last one folds the conversion to the standard form (and the 1/N of a plain inverse
        // transform) into its inter-pass twiddles, read from a standard-form table (ntt.hip NTT_FOLD)
        if (!inverse) rc = ctx_get_twiddles(c, log_n, &job.tw_last);
        else if (!coset) {
            rc = ctx_get_twiddles_ninv(c, log_n, &job.tw_last);
            job.tw_last_has_post = 1;
        }
        if (rc) return rc;
    }
    if (!inverse && coset) {  // coeff_to_extended: a_i *= zeta^(i mod 3)
        job.has_pre = 1;
        job.pre[0] = Fr::one();
        job.pre[1] = c->zeta;
        job.pre[2] = c->zeta2;
    }
    if (inverse) {  // x 1/N, and for the coset also zeta^-(i mod 3) = {1, zeta^2, zeta}
        // 1 / N: a constant of the transform size, inverted once per context (19 us of Fermat on the launching thread per
        // inverse transform before)
        auto nit = c->ninv.find(log_n);
        if (nit == c->ninv.end()) nit = c->ninv.emplace(log_n, fe_inv_fast(fr_from_u64(N))).first;
        const Fr ninv = nit->second;
        job.has_post = 1;
        job.post[0] = ninv;
        job.post[1] = coset ? fe_mul(ninv, c->zeta2) : ninv;
        job.post[2] = coset ? fe_mul(ninv, c->zeta) : ninv;
    }
    if (c->audit.on) {  // reads the sources, writes the destinations, ping-pongs through the context's ONE scratch
        const void *rd[NTT_MAX_BATCH + 1], *wr[NTT_MAX_BATCH + 1];
        for (uint32_t b = 0; b < batch; b++) {
            rd[b] = srcs[b];
            wr[b] = dsts[b];
        }
        rd[batch] = wr[batch] = c->scratch;
        c->audit.op_v(st, rd, batch + 1, wr, batch + 1, "NTT batch");
    }
    aud_record(c, c->ev[ZK_T_NTT][0], st);
    hipError_t e = ntt_run(job, st);
    aud_record(c, c->ev[ZK_T_NTT][1], st);
    c->ev_valid[ZK_T_NTT] = true;
    if (e != hipSuccess) {
        c->last_hip = (int)e;
        return ZK_EHIP;
    }
    return ZK_OK;
}

// ---- the three-coset route (poly.hip "three cosets") ---------------------------------------------------------------
static int ctx_get_coset3_pre(zk_ctx* c, uint32_t k, const Fr** out) {
    auto it = c->coset3_pre.find(k);
    if (it != c->coset3_pre.end()) {
        *out = it->second;
        return ZK_OK;
    }
    const Fr* tw_ext = nullptr;
    int rc = ctx_get_twiddles(c, k + 2, &tw_ext);
    if (rc) return rc;
    const size_t n = (size_t)1 << k;
    Fr* tab = nullptr;
    if (hipMalloc(&tab, 2 * n * sizeof(Fr)) != hipSuccess) return ZK_ENOMEM;
    const Fr k1024 = fr_from_u64(1024);
    const Fr zp[3] = {k1024, fe_mul(c->zeta, k1024), fe_mul(c->zeta2, k1024)};
    for (uint32_t j = 1; j <= 2; j++) launch_coset3_pre(tw_ext, (uint32_t)n, j, zp, tab + (size_t)(j - 1) * n, c->stream);
    aud_sync(c, c->stream);  // made once per context and size; read from any of the context's streams afterwards
    c->coset3_pre[k] = tab;
    *out = tab;
    return ZK_OK;
}

int ctx_ntt_cosets3(zk_ctx* c, const Fr* const* polys, Fr* const* dsts, uint32_t cols, uint32_t k, hipStream_t on) {
    const hipStream_t st = on ? on : c->stream;
    const size_t n = (size_t)1 << k;
    const uint32_t batch = 3 * cols;
    if (cols == 0 || batch > ctx_ntt_max_batch(k)) return ZK_EINVAL;
    int rc = ctx_ensure_scratch(c, n * batch);
    if (rc) return rc;
    const Fr *tw = nullptr, *pre = nullptr;
    if ((rc = ctx_get_twiddles_ntt(c, k, &tw)) != ZK_OK || (rc = ctx_get_coset3_pre(c, k, &pre)) != ZK_OK) return rc;
    NttJob job;
    memset(&job, 0, sizeof(job));
    job.batch = batch;
    for (uint32_t q = 0; q < cols; q++)
        for (uint32_t j = 0; j < 3; j++) {
            job.srcs[3 * q + j] = polys[q];
            job.dsts[3 * q + j] = dsts[q] + (size_t)j * n;
            job.pre_tabs[3 * q + j] = j ? pre + (size_t)(j - 1) * n : nullptr;  // coset 0: zeta^m alone (`pre`)
        }
    job.tmp = c->scratch;
    job.tw = tw;
    job.log_n = k;
    job.n_in = job.n_out = (uint32_t)n;
    job.max_log_r = c->opt_ntt_max_r;
    if (k > 7 && (rc = ctx_get_twiddles(c, k, &job.tw_last)) != ZK_OK) return rc;
    job.has_pre = 1;
    job.pre[0] = Fr::one();
    job.pre[1] = c->zeta;
    job.pre[2] = c->zeta2;
    if (c->audit.on) {
        const void *rd[NTT_MAX_BATCH + 1], *wr[NTT_MAX_BATCH + 1];
        for (uint32_t q = 0; q < cols; q++) {
            rd[q] = polys[q];
            wr[q] = dsts[q];
        }
        rd[cols] = wr[cols] = c->scratch;
        c->audit.op_v(st, rd, cols + 1, wr, cols + 1, "NTT batch (three cosets)");
    }
    aud_record(c, c->ev[ZK_T_NTT][0], st);
    hipError_t e = ntt_run(job, st);
    aud_record(c, c->ev[ZK_T_NTT][1], st);
    c->ev_valid[ZK_T_NTT] = true;
    if (e != hipSuccess) {
        c->last_hip = (int)e;
        return ZK_EHIP;
    }
    return ZK_OK;
}

int ctx_intt_cosets3(zk_ctx* c, Fr* h, uint32_t k) {
    const size_t n = (size_t)1 << k;
    const Fr* srcs[3] = {h, h + n, h + 2 * n};
    Fr* dsts[3] = {h, h + n, h + 2 * n};
    int rc = ctx_ntt_batch(c, srcs, n, dsts, 3, k, true, false, n);  // plain inverse transforms (1/n included), in place
    if (rc) return rc;
    const Fr* tw_ext = nullptr;
    if ((rc = ctx_get_twiddles(c, k + 2, &tw_ext)) != ZK_OK) return rc;
    auto it = c->coset3_consts.find(k);
    if (it == c->coset3_consts.end()) {
        const Fr z = fe_pow_u64(c->zeta, n), i4 = fe_pow_u64(fr_omega(k + 2), n);
        const Fr two = fr_from_u64(2);
        Coset3Consts q;
        q.inv2 = fe_inv_fast(two);
        q.inv_2z = fe_inv_fast(fe_mul(two, z));
        q.zi = fe_mul(z, i4);
        q.inv_2z2 = fe_inv_fast(fe_mul(two, fe_sqr(z)));
        q.zinv[0] = Fr::one();
        q.zinv[1] = c->zeta2;  // zeta^-1 = zeta^2 (a cube root of unity)
        q.zinv[2] = c->zeta;
        it = c->coset3_consts.emplace(k, q).first;
    }
    if (c->audit.on) c->audit.op(c->stream, {h}, {h}, "three cosets -> h pieces");
    launch_coset3_combine(h, tw_ext, (uint32_t)n, it->second, c->stream);
    return ZK_OK;
}

static int ntt_resident(zk_ctx* c, PolyRec* src, PolyRec* dst, uint32_t log_n, bool inverse, bool coset, size_t n_out) {
    return ctx_ntt(c, src->ptr, src->n, dst->ptr, log_n, inverse, coset, n_out);
}

static uint32_t log2_exact(size_t n) {
    uint32_t l = 0;
    while (((size_t)1 << l) < n) l++;
    return ((size_t)1 << l) == n ? l : 0xffffffffu;
}

ZK_API(zk_lagrange_to_coeff, (zk_ctx* c, zk_poly h), (c, h)) {
    if (!c) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    PolyRec* r = find_poly(c, h);
    if (!r) return ZK_EINVAL;
    const uint32_t lg = log2_exact(r->n);
    if (lg > 26) return ZK_EINVAL;
    int rc = ctx_bind(c);
    if (rc) return rc;
    return ntt_resident(c, r, r, lg, true, false, r->n);
}

ZK_API(zk_coeff_to_lagrange, (zk_ctx* c, zk_poly h), (c, h)) {
    if (!c) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    PolyRec* r = find_poly(c, h);
    if (!r) return ZK_EINVAL;
    const uint32_t lg = log2_exact(r->n);
    if (lg > 26) return ZK_EINVAL;
    int rc = ctx_bind(c);
    if (rc) return rc;
    return ntt_resident(c, r, r, lg, false, false, r->n);
}

ZK_API(zk_coeff_to_extended, (zk_ctx* c, zk_poly src, zk_poly dst), (c, src, dst)) {
    if (!c) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    PolyRec *s = find_poly(c, src), *d = find_poly(c, dst);
    if (!s || !d || s == d) return ZK_EINVAL;
    const uint32_t lg = log2_exact(d->n);
    if (lg > 26 || s->n > d->n) return ZK_EINVAL;
    int rc = ctx_bind(c);
    if (rc) return rc;
    return ntt_resident(c, s, d, lg, false, true, d->n);
}

ZK_API(zk_extended_to_coeff, (zk_ctx* c, zk_poly ext, size_t n_out), (c, ext, n_out)) {
    if (!c) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    PolyRec* r = find_poly(c, ext);
    if (!r || n_out > r->n) return ZK_EINVAL;
    const uint32_t lg = log2_exact(r->n);
    if (lg > 26) return ZK_EINVAL;
    int rc = ctx_bind(c);
    if (rc) return rc;
    return ntt_resident(c, r, r, lg, true, true, n_out);
}

ZK_API(zk_eval, (zk_ctx* c, zk_poly h, const uint64_t x[4], uint64_t out[4]), (c, h, x, out)) {
    if (!c || !x || !out) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    PolyRec* r = find_poly(c, h);
    if (!r || r->n > 0xffffffffu) return ZK_EINVAL;
    int rc = ctx_bind(c);
    if (rc) return rc;
    Fr xx;
    memcpy(&xx, x, 32);
    const uint32_t blocks = eval_blocks((uint32_t)r->n);
    aud_record(c, c->ev[ZK_T_EVAL][0], c->stream);
    launch_eval(r->ptr, (uint32_t)r->n, xx, c->small, c->stream);
    aud_record(c, c->ev[ZK_T_EVAL][1], c->stream);
    c->ev_valid[ZK_T_EVAL] = true;
    HIPCHK(c, hipMemcpyAsync(c->host_small, c->small + blocks, sizeof(Fr), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, aud_sync(c, c->stream));
    memcpy(out, c->host_small, 32);
    return ZK_OK;
}

namespace zk {
uint32_t kate_division_scratch(uint32_t n);  // prover_kernels.hip
void launch_kate_division(const Fr* p, Fr* q, uint32_t n, const Fr& z, Fr* scratch, hipStream_t st);
}  // namespace zk

ZK_API(zk_kate_division, (zk_ctx* c, zk_poly hp, const uint64_t z[4], zk_poly hq), (c, hp, z, hq)) {
    if (!c || !z) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    PolyRec* p = find_poly(c, hp);
    PolyRec* q = find_poly(c, hq);
    if (!p || !q || p->n != q->n || p->n == 0 || p->n > ((size_t)1 << 26)) return ZK_EINVAL;
    int rc = ctx_bind(c);
    if (rc) return rc;
    if ((rc = ctx_ensure_scratch(c, kate_division_scratch((uint32_t)p->n)))) return rc;
    Fr zz;
    memcpy(&zz, z, 32);
    launch_kate_division(p->ptr, q->ptr, (uint32_t)p->n, zz, c->scratch, c->stream);
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, aud_sync(c, c->stream));
    return ZK_OK;
}


