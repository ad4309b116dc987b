// ctx.h — the engine context shared by engine.hip (C-ABI) and prover.hip.
#pragma once
#include <map>
#include <memory>
#include <new>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "../../include/zkmi355.h"
#include "engine.h"
#include "audit.h"
#include "hostutil.h"

using namespace zk;

struct zk_pk_rec;

struct PolyRec {
    Fr* ptr;
    size_t n;
};

// The resident SRS of a context: both bases and their window tables.  Reference-counted: contexts made with
// zk_ctx_create_shared use the block of the context they were made from (read-only; several proof pipelines on one device
// then gather from ONE copy of the tables), and a context that loads another SRS simply lets go of its reference.
struct SrsBlock {
    int device = -1;
    G1Affine *g = nullptr, *g_lagrange = nullptr, *g_table = nullptr, *g_lagrange_table = nullptr;
    ~SrsBlock() {
        if (device >= 0) hipSetDevice(device);
        if (g) hipFree(g);
        if (g_lagrange) hipFree(g_lagrange);
        if (g_table) hipFree(g_table);
        if (g_lagrange_table) hipFree(g_lagrange_table);
    }
};

struct zk_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    int stream_slot = -1;              // the context's slot in its device's stream pool (engine.hip): main and side streams are the slot's; -1: its own
    bool stream_own_priority = false;  // the main stream was made by ZK_OPT_STREAM_PRIORITY: the context's own, destroyed with it
    int last_hip = 0;
    StreamAudit audit;  // ZK_OPT_STREAM_AUDIT (audit.h): the happens-before ledger of this context's streams, off by default
    bool audit_fault = false;  // ZK_OPT_STREAM_AUDIT = 2: the audit's self-test (the prover takes a knowingly unordered path)
    std::mutex mu;
    std::map<uint32_t, Fr*> twiddles;      // log_n -> w_{2^log_n}^i table (standard form: quotient, permutation kernels)
    std::map<uint32_t, Fr*> coset_points;  // zeta * w^i (standard form): the x of the quotient's permutation terms
    std::map<uint32_t, Fr*> twiddles_ntt;  // the same powers in the NTT's internal form (x 2^261, ntt.hip)
    std::map<uint32_t, Fr> ninv;           // 1 / 2^log_n (host constant of the inverse transforms)
    std::map<uint32_t, Fr*> twiddles_ninv; // w^i / 2^log_n (standard form): the last pass of an inverse transform (ntt.hip NTT_FOLD)
    std::map<uint32_t, Fr*> coset3_pre;    // k -> [2][2^k]: the twists (zeta w_4n^j)^m, j = 1, 2, of the three-coset transforms (poly.hip)
    std::map<uint32_t, Coset3Consts> coset3_consts;  // k -> the constants of the 3 x 3 solve
    uint32_t opt_quotient_domain = 0;      // ZK_OPT_QUOTIENT_DOMAIN: 0 auto (three cosets from k = 16), 1 always the whole extended domain, 2 three cosets wherever h has three pieces
    // SRS (the pointers below alias the members of `srs`, the owner)
    std::shared_ptr<SrsBlock> srs;
    int srs_k = -1;
    G1Affine* g = nullptr;
    G1Affine* g_lagrange = nullptr;
    G1Affine* g_table = nullptr;           // window multiples of g / g_lagrange (fixed-base MSM)
    G1Affine* g_lagrange_table = nullptr;
    uint32_t table_c = 0;
    bool g_has_identity = true, g_lagrange_has_identity = true;  // set with the tables (msm_bases_have_identity)
    // G2 half of ParamsKZG (g2, s_g2 = [s]G2) as raw Montgomery images x.c0 || x.c1 || y.c0 || y.c1: the engine never
    // computes with it, it only travels through the SRS file (zk_srs_write / zk_srs_read)
    bool g2_valid = false;
    uint8_t g2_raw[128] = {0}, s_g2_raw[128] = {0};
    uint64_t srs_gen = 0;  // bumped by every zk_srs_setup / zk_srs_load / zk_srs_read: keys remember the SRS they were made under
    // tuning options (zk_ctx_set_option); 0 = built-in choice
    uint32_t opt_msm_window = 0, opt_msm_batch = 0, opt_ntt_max_r = 0, opt_gp_batch_invert = 0;
    uint32_t opt_tail_stream = 0;  // ZK_OPT_MSM_TAIL_STREAM: 0 auto, 1 always the context's tail stream, 2 always the main stream
    uint32_t opt_tail_main_above = 0;  // ZK_OPT_MSM_TAIL_MAIN_ABOVE: auto mode puts the tails on the main stream with MORE than this many contexts active on the device (0 = the measured default, 2)
    uint32_t opt_batch_pass_cols = 0;  // ZK_OPT_BATCH_PASS_COLUMNS: columns per MSM pass of a lock-step batch (0 = max(min(2 B, 8), the single prover's pass width))
    uint32_t msm_min_cols = 0;         // the lanes' fixed-base workspaces take at least this many columns per pass (raised by zk_prove_batch, never lowered: the wider workspaces are kept)
    uint32_t opt_no_activity_hold = 0;  // ZK_OPT_ACTIVITY_HOLD = 1
    bool opt_activity_pinned = false;   // ZK_OPT_ACTIVITY_HOLD = 2: active until the option is changed
    bool act_held = false;         // inside a whole-proof call: the slot counts as active whatever its last stamp (ctx_activity_hold)
    int act_slot = -1;             // this context's slot in its device's activity table (engine.hip ctx_activity_*)
    // The context's TAIL stream (round 4: one, shared by the lanes; rounds 2-3 had one per lane).  Where a pass's reduction tail
    // runs is decided per pass (ctx_msm_begin_batch): on this stream while at most two contexts are ACTIVE on the device (have
    // enqueued an MSM pass within the last 4 ms, through whatever entry point: ctx_activity_touch) — a lone proof hides its
    // tails behind its next head —, on the main stream beyond that: HIP maps a process's streams onto FOUR hardware
    // queues, and streams that share a queue run in order, so with three or more pipelines every extra stream puts some
    // pipeline's kernels behind another's accumulation (measured, tools/queue_ab2.sh: 4 pipelines 103.7 proofs/s with one
    // stream each, 99.1 with a tail stream each even on 8 queues, 94.4 for round 3's 2 pipelines x 4 streams)
    hipStream_t tail_stream = nullptr;
    // The context's TRANSFORM stream (round 5): the coefficient / extended-coset forms of a proof's columns are needed by the
    // quotient only, so a LONE proof (no other context active on the device: three streams fit the four hardware queues)
    // runs those NTTs beside its MSM passes instead of between them — both are bound by the issue of their own instruction
    // streams and leave each other's stalls to fill.  ev_rows: the last flush of staged blinding rows on the main stream (what
    // a column's transforms wait for); ev_xform: the last transform enqueued (what the quotient waits for).
    // The context's MSM stream (round 5, experiment ZK_OPT_MSM_STREAM): a lone proof's MSM passes (sort head + accumulation) run
    // here, behind an event taken on the main stream when the pass is begun (its inputs are complete), so that the glue kernels
    // of the next phase — lookup permutation, grand products with their host round trip — do not queue behind an accumulation
    hipStream_t msm_stream = nullptr;
    hipEvent_t ev_msm_in = nullptr;
    bool msm_side = false;          // set by the prover for the duration of a proof
    uint32_t opt_msm_stream = 0;    // ZK_OPT_MSM_STREAM: 0 auto, 1 side stream, 2 main stream
    uint32_t opt_msm_t1 = 0;        // ZK_OPT_MSM_T1: 1 one lane per bucket, 0 / 2 parts + segmented tree (default) (msm.hip msm_wbucket_kernel)
    hipStream_t xform_stream = nullptr;
    hipEvent_t ev_rows = nullptr, ev_xform = nullptr;
    bool xform_pending = false;
    uint32_t opt_xform_stream = 0;  // ZK_OPT_XFORM_STREAM: 0 auto (side stream for a lone context), 1 side stream, 2 main stream
    // MSM lanes: each in-flight MSM owns a workspace and a pinned result buffer
    static constexpr int MSM_LANES = 3;
    struct MsmLane {
        MsmWorkspace* ws = nullptr;      // fixed-base mode over the resident SRS (window = the tables')
        MsmWorkspace* ws_gen = nullptr;  // arbitrary bases (the fine-grained seam): its own workspace, so that a host mixing both does not rebuild one per call
        MsmWorkspace* ws_run = nullptr;  // the one the MSM in flight uses
        hipStream_t tail = nullptr;      // the stream the tail of the MSM in flight runs on (the context's tail stream or its main stream)
        hipEvent_t head_done = nullptr, tail_done = nullptr;
        hipEvent_t t_head[2] = {nullptr, nullptr};  // timing: the pass as enqueued on the main stream
        hipEvent_t t_acc[4] = {nullptr, nullptr, nullptr, nullptr};  // timing: accumulate kernel [0, 1]; reduction tail [2, 3] (wide path)
        size_t n = 0;
        const G1Affine* table = nullptr;  // the window table of the MSM in flight (fixed-base mode)
        G1X* host_buf = nullptr;  // pinned
        bool busy = false;
        uint32_t nwin = 0, cw = 0;
        uint32_t batch = 1;       // columns of the MSM in flight (fixed-base mode), results collected together
        bool fixed = false;       // the MSM in flight runs over the resident SRS's window tables
    } lanes[MSM_LANES];
    // scratch
    Fr* scratch = nullptr;
    size_t scratch_n = 0;
    // device staging of the fine-grained seam (zk_msm_bn254 / zk_ntt_bn254_fr): grow-only, reused across calls
    void* seam_buf[2] = {nullptr, nullptr};
    size_t seam_bytes[2] = {0, 0};
    Fr* small = nullptr;  // 2048 + 8 elements for reductions
    Fr* host_small = nullptr;  // pinned, 8 elements
    // polys
    std::unordered_map<uint64_t, PolyRec> polys;
    // vectors given back by zk_poly_free, handed out again by zk_poly_alloc for the same length: hipFree waits for the whole
    // device (for every other context's kernels too), so a host that allocates and frees its request's columns around every
    // proof would stall all pipelines of the GPU per request.  At most POLY_SPARE_MAX vectors / POLY_SPARE_BYTES bytes.
    static constexpr size_t POLY_SPARE_MAX = 16, POLY_SPARE_BYTES = (size_t)2 << 30;
    std::vector<PolyRec> poly_spare;
    size_t poly_spare_bytes = 0;
    uint64_t next_handle = 1;
    // constants
    Fr zeta, zeta2;
    // proving keys
    std::unordered_map<uint64_t, zk_pk_rec*> pks;
    // timing
    hipEvent_t ev[ZK_T_COUNT][2] = {};
    bool ev_valid[ZK_T_COUNT] = {false};
    double acc_ms[ZK_T_COUNT] = {0};   // accumulated over calls since zk_timer_reset (MSM kinds only)
    uint64_t acc_n[ZK_T_COUNT] = {0};
    uint64_t msm_launches = 0;
    float last_plain_ms[ZK_T_COUNT] = {0};
};

// Every `extern "C"` entry point is defined through ZK_API: the body runs inside a try block, so that no C++
// exception (std::bad_alloc from the host-side containers, anything else) crosses the C boundary.
// the cross-stream calls of the engine go through these: the HIP call, and the ledger's twin of it when the audit is on
static inline hipError_t aud_record(zk_ctx* c, hipEvent_t ev, hipStream_t st) {
    const hipError_t e = hipEventRecord(ev, st);
    c->audit.record(ev, st);
    return e;
}
static inline hipError_t aud_wait(zk_ctx* c, hipStream_t st, hipEvent_t ev) {
    const hipError_t e = hipStreamWaitEvent(st, ev, 0);
    c->audit.wait(st, ev);
    return e;
}
static inline hipError_t aud_sync(zk_ctx* c, hipStream_t st) {
    const hipError_t e = hipStreamSynchronize(st);
    if (e == hipSuccess) c->audit.host_stream(st);
    return e;
}
static inline hipError_t aud_esync(zk_ctx* c, hipEvent_t ev) {
    const hipError_t e = hipEventSynchronize(ev);
    if (e == hipSuccess) c->audit.host_event(ev);
    return e;
}
// an entry point's verdict under the audit: a violation recorded while it ran turns ZK_OK into ZK_EINTERNAL
static inline int aud_verdict(zk_ctx* c, uint64_t violations_before, int rc) {
    return (rc == ZK_OK && c->audit.on && c->audit.violations != violations_before) ? ZK_EINTERNAL : rc;
}

#define ZK_API(name, params, args)                          \
    static int name##_impl params;                          \
    extern "C" int name params {                            \
        try {                                               \
            return name##_impl args;                        \
        } catch (const std::bad_alloc&) {                   \
            return ZK_ENOMEM;                               \
        } catch (...) {                                     \
            return ZK_EINTERNAL;                            \
        }                                                   \
    }                                                       \
    static int name##_impl params

#define HIPCHK(ctx, x)                 \
    do {                               \
        hipError_t _e = (x);           \
        if (_e != hipSuccess) {        \
            (ctx)->last_hip = (int)_e; \
            return ZK_EHIP;            \
        }                              \
    } while (0)


int ctx_bind(zk_ctx* c);
// contexts active on a device (all contexts of THIS process that enqueued an MSM pass within the last few ms): decides where
// the MSM reduction tails run (engine.hip)
void ctx_activity_register(zk_ctx* c);
void ctx_activity_unregister(zk_ctx* c);
int ctx_activity_touch(zk_ctx* c);
void ctx_activity_hold(zk_ctx* c, bool on);
int ctx_lone_streams(zk_ctx* c);  // creates the transform / MSM streams of a lone proof on first use
void ctx_release_spares(zk_ctx* c);  // frees the vectors zk_poly_free parked (caller holds c->mu, device bound)
int ctx_ensure_scratch(zk_ctx* c, size_t n);
int ctx_get_twiddles(zk_ctx* c, uint32_t log_n, const Fr** out);
int ctx_get_twiddles_ntt(zk_ctx* c, uint32_t log_n, const Fr** out);
int ctx_get_twiddles_ninv(zk_ctx* c, uint32_t log_n, const Fr** out);
int ctx_get_coset_points(zk_ctx* c, uint32_t log_n, const Fr** out);  // zeta * w^i: the points of the extended coset
// MSM of device-resident scalars against device-resident bases -> Jacobian on host (synchronises)
int ctx_msm_device(zk_ctx* c, const Fr* d_scalars, const G1Affine* d_bases, size_t n, G1Jac* out);
// split form: begin enqueues the MSM on lane `lane` (head on the context stream, tail on the
// lane's stream) and returns; end waits for the lane and finishes on the host.
int ctx_msm_begin(zk_ctx* c, int lane, const Fr* d_scalars, const G1Affine* d_bases, size_t n);
int ctx_msm_end(zk_ctx* c, int lane, G1Jac* out);
// several columns against the same resident SRS basis in one pass (at most ctx_msm_max_batch(c) of them);
// ctx_msm_end_batch writes one result per column, in order
int ctx_msm_begin_batch(zk_ctx* c, int lane, const Fr* const* d_scalars, uint32_t batch, const G1Affine* d_bases, size_t n);
int ctx_msm_end_batch(zk_ctx* c, int lane, G1Jac* out);
uint32_t ctx_msm_max_batch(const zk_ctx* c);
void ctx_msm_drain(zk_ctx* c);  // error paths: wait for every MSM in flight and drop its result
// NTT between device buffers: inverse => x 1/N; coset => zeta scaling (coeff_to_extended / extended_to_coeff)
int ctx_ntt(zk_ctx* c, const Fr* src, size_t src_n, Fr* dst, uint32_t log_n, bool inverse, bool coset, size_t n_out);
// the same transform over `batch` vectors in one launch per pass (batch <= ctx_ntt_max_batch(log_n))
// the three-coset route (poly.hip "three cosets"): coefficient vectors (n = 2^k) -> [3][n] coset-major values, `cols` columns per
// call (3 cols <= ctx_ntt_max_batch(k)); and the quotient's way back, in place: [3][n] values of h -> its 3n coefficients
int ctx_ntt_cosets3(zk_ctx* c, const Fr* const* polys, Fr* const* dsts, uint32_t cols, uint32_t k, hipStream_t on = nullptr);
int ctx_intt_cosets3(zk_ctx* c, Fr* h, uint32_t k);
int ctx_ntt_batch(zk_ctx* c, const Fr* const* srcs, size_t src_n, Fr* const* dsts, uint32_t batch, uint32_t log_n, bool inverse,
                  bool coset, size_t n_out, hipStream_t on = nullptr /* the context's main stream */);
uint32_t ctx_ntt_max_batch(uint32_t log_n);
void pk_destroy_all(zk_ctx* c);
// SRS plumbing shared by engine.hip (setup / load) and serde.hip (read)
int srs_alloc(zk_ctx* c, uint32_t k);
// replaces the resident SRS by two decoded, validated bases of 2^k points (device buffers the context takes over);
// the previous SRS, its window tables and every key made under it are dropped only now
void srs_adopt(zk_ctx* c, uint32_t k, G1Affine* g, G1Affine* g_lagrange);
int srs_build_tables(zk_ctx* c, uint32_t k);
void srs_set_g2_from_secret(zk_ctx* c, const Fr& s_mont);
