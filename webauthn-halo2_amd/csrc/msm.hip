// msm.hip — Pippenger bucket multi-scalar multiplication over BN254 G1 for gfx950.
//
// Device replacement for halo2_proofs `arithmetic::best_multiexp`, reached only
// through `ParamsKZG::{commit, commit_lagrange}` (reference
// call sites halo2-circuits/src/ecc/ecdsa_p256.rs:366-373,416-423,555-562).
// Any correct algorithm yields the same group element, so the result is
// bit-identical to the reference's after affine normalisation.
//
// Layout of the source (round 6; one translation unit — the parts are textual includes that share the constants and the
// workspace record defined here):
//   msm.hip             constants, the workspace, the choice of the window plan, msm_run / msm_run_wide (the launch sequences),
//                       the host's finish (Horner over the bit sums)
//   msm_wide.hip.h      THE path of every proof of the reference's configurations (2^16 .. 2^21 points over the resident SRS):
//                       15 .. 17-bit windows on precomputed window tables, ONE bucket set per column, a dense two-level sort,
//                       restartable accumulation lanes, the three-kernel reduction tail
//   msm_legacy.hip.h    the plans around it: lg n - 5 window bits with digit planes and padded segments below 2^16 points, the
//                       swept sort from 2^22, and arbitrary bases (`zk_msm_bn254`: W windows x 2^(c-1) buckets, host Horner over
//                       the windows) — kernels msm_recode / msm_sort / msm_scatter* / msm_accumulate / msm_gather* / msm_bitsum
//   msm_tables.hip.h    the window tables 2^(c w) P_i and the identity test of a basis
// Two modes:
//   generic   bases are arbitrary (the fine-grained `zk_msm_bn254` seam): W windows of c bits, W x 2^(c-1) buckets;
//   fixed     bases are the resident SRS (`zk_commit`): a table of window multiples 2^(c w) P_i is precomputed once per basis,
//             so every window's digits fall into ONE set of 2^(c-1) buckets — no per-window reduction, no window Horner.
// Load balance does not depend on the scalar distribution: witness columns are dominated by zeros / small values (hot low
// buckets), and an accumulation lane is a fixed number of entries whatever buckets they fall in.
#include <stdlib.h>
#include <string.h>
#include <vector>

// This file's products on the carry-free field keep the source order of their multiply-adds (field29.hip.h ZK_MUL29_ASM):
// the accumulation runs four waves per SIMD (below), enough to hide a serial column, and saves the 64-bit join per column.
// Round 6: = 2, a column part per asm statement instead of a multiply-add per statement — hipcc's hazard recogniser puts an
// s_nop behind every asm statement whose result the next instruction reads (1 358 per mixed addition, 283 now): the lone 2^19
// accumulation 0.572 -> 0.553 / 0.580 -> 0.566 ms, two columns 1.058 -> 1.040 / 1.069 -> 1.033 (profiles/r6_ab_block_asm.txt)
#ifndef ZK_MUL29_ASM
#define ZK_MUL29_ASM 2
#endif
// ... except in the reduction kernels, which run about one wave per SIMD: their additions wait on that serial chain, not on
// issue slots, and take the compiler's form (g1x29_add<false>; tools/tail_times.sh, per lone k = 19 proof: msm_wrowcol
// 1.19 -> 0.98 ms, msm_wbits 1.00 -> 0.59, k = 15: msm_bitsum 1.02 -> 0.75, msm_gather 0.75 -> 0.64; msm_wparts unchanged)
#ifndef ZK_TAIL_SER
#define ZK_TAIL_SER false
#endif
// ... and except the per-bucket T1 (msm_wbucket_kernel), which is chosen only where the chip is loaded: issue slots count there
#ifndef ZK_T1B_SER
#define ZK_T1B_SER true
#endif
// slots a bucket may have for the per-bucket T1 to take it (a uniformly random 2^19 column: 17 .. 19)
#ifndef ZK_T1B_LMAX
#define ZK_T1B_LMAX 24
#endif
#include "ec29.hip.h"
#include "engine.h"

namespace zk {

static constexpr uint32_t SIGN_BIT = 0x80000000u;
static constexpr uint32_t SKIP_ENTRY = 0xffffffffu;  // padding entry (no base)
// wide path: an entry is sign << 31 | first-of-bucket << 30 | delta << 24 | table index (24 bits)
static constexpr uint32_t WIDE_FLAG = 0x40000000u;  // index bits ib = 24 .. 26 (WideGeo): delta field [ib, 30), escape = all ones
static constexpr uint32_t CHUNK = 16384;  // scalars per histogram / scatter workgroup
static constexpr uint32_t SORT_LDS_BUCKETS = 8192;  // 32 KiB of LDS counters per sort workgroup
#ifndef ZK_SEG0  // build-time tuning knobs (tools/ab_variants.sh)
#define ZK_SEG0 16
#endif
#ifndef ZK_GA
#define ZK_GA 4
#endif
static constexpr uint32_t SEG0 = ZK_SEG0;  // entries per accumulate lane
static constexpr uint32_t GA = ZK_GA;      // slots per first-level gather lane
#ifndef ZK_GLANES
#define ZK_GLANES 16
#endif
static constexpr uint32_t GLANES_FIXED = ZK_GLANES;  // lanes per bucket in the second-level gather (fixed-base mode)
#ifndef ZK_GSHARE
#define ZK_GSHARE 64
#endif
static constexpr uint32_t GSHARE = ZK_GSHARE;         // first-level partials per (bucket, part) group; a bucket uses ceil(partials / GSHARE) parts
static constexpr uint32_t PAD = SEG0 * GA;  // bucket ranges are padded to multiples of PAD entries

struct MsmBatch {
    const Fr* s[MSM_MAX_BATCH];
};

struct MsmWorkspace {
    size_t max_n;
    uint32_t max_batch;         // columns per fixed-base launch this workspace is sized for
    uint32_t c, nwin, nb;       // window bits, windows, buckets per window
    uint32_t parts_fixed, parts_generic;
    size_t slot_elems;          // elements of slot_pt
    int16_t* digits;            // [nwin][max_n]  (|digit| <= 2^(c-1) <= 8192)
    uint32_t* totals;           // [nwin*nb + 1]
    uint32_t* bucket_start;     // [nwin*nb + 1]
    uint32_t* blockbase;        // [nblk][nb]
    uint32_t* counts;           // [4]
    uint32_t* entries;          // [max_n * nwin + padding]: +-base index, bucket implied by position
    // two-level sort (fixed-base mode): entries first grouped by coarse bin (bucket / 64) in `inter`, then by bucket
    uint32_t* inter;            // [max_batch][max_n * nwin]
    uint32_t* coarse;           // per column: COARSE_WORDS (coarse-bin starts, chunk prefix, append cursors), then [blocks][CBINS_MAX] reserved bases
    uint32_t coarse_stride;     // words per column in `coarse`
    uint32_t* cursor;           // [max_batch * nb] per-bucket write cursors of the second level
    size_t inter_stride;        // entries per column in `inter`
    uint32_t* redo;             // [entries / SEG0] segments the unchecked accumulation hands to the checked one
    G1X29S* slot_pt;            // [entries / SEG0]  partial sums stay in the accumulation's internal form (ec29.hip.h)
    G1X29S* partial;            // [entries / PAD]
    G1X29S* part;                  // [nbt * parts]
    G1X* bit_sum;               // [nwin * c]
    // wide path (15 / 16-bit windows, fixed-base mode): per-column regions
    bool wide;
    bool w_redo_valid;          // the last pass on this workspace ran the wide path's unchecked accumulation over n > 0 scalars: the word
                                // behind its sums in the host buffer is that pass's redo count (msm_wide_redo_count); any other pass —
                                // the standard plan on a wide workspace, an empty one — leaves no such word
    uint32_t w_t1_mode;         // T1 of the wide path (ZK_OPT_MSM_T1): 1 one lane per bucket, otherwise parts + segmented tree
    bool w_last_wide;           // the last pass on this workspace took the wide path (its tail's timing events were recorded)
    bool w_clean;               // the pass counters (totals, cursors, counts) are zero: the previous wide pass left them so
    // per-bucket totals and level-2 cursors exist twice: pass i counts in set i & 1 while its first kernel — 131 K lanes with
    // to spare (the fine histogram: two thousand workgroups) — zeroes the other set for pass i + 1 (in the 16 waves of the last
    // tail kernel the reset cost 25 us of exposed latency per pass, in H1 13 us)
    uint32_t* w_tot[2];
    uint32_t* w_cur[2];
    uint32_t w_par;
    uint32_t w_used_cols[2];    // columns of a set that hold counts of its last pass (what the next H1 has to zero)
    uint32_t w_ib, w_fb;        // entry layout: table index bits, fine key bits (msm_wide_shape)
    size_t w_ent_stride;        // entries per column region (multiple of 64)
    uint32_t w_lane_stride;     // accumulation lanes per column region
    uint32_t w_slot_stride;     // slots per column region: lanes + buckets
    uint32_t w_part_stride;     // parts per column region
    uint32_t* w_lane_b;         // [max_batch][w_lane_stride] the bucket every lane starts in
    uint32_t* w_bstart;         // [max_batch][nb] places of the buckets in their column's entry region
    uint32_t* w_pstart;         // [max_batch][nb] first part of every bucket
    uint32_t* w_pbucket;        // [max_batch][w_part_stride] bucket of every part
    G1X29S* w_part;             // [max_batch][w_part_stride]
    G1X29S* w_rc;               // [max_batch][nb / 256 + 256] row and column sums of the bucket matrix
};

static inline uint32_t nwin_for(uint32_t c) { return 254 / c + 1; }
// smallest log2(n) that takes 17-bit windows by default.  99 = never: measured end to end (round 4, tools/bench_ab.sh, same box,
// new head in both): 17 bits make the accumulation 6 % shorter (0.627 against 0.666 ms per launch: 15 additions per scalar
// instead of 16) and the proof NOT faster — 94.2 / 94.7 proofs/s against 95.7 / 95.5 at 16 bits, single proof 12.3 against 12.0 ms:
// twice the buckets (65 536) double the row / column sums and the parts of the reduction tail, which run at one or two waves per
// SIMD.  The 17-bit plan stays selectable (ZK_OPT_MSM_WINDOW = 17) and tested.
#ifndef ZK_W17_MIN_LG
#define ZK_W17_MIN_LG 99
#endif
// the wide path (below): 15 / 16 / 17-bit windows over a resident basis.  An entry holds the table index in ib = 24 .. 26 bits
// and a fine key of fb = 31 - ib bits (at most 7); the buckets / 2^fb coarse bins must not exceed 512
static bool msm_wide_shape(uint32_t c, size_t table_stride, uint32_t* ib_out, uint32_t* fb_out) {
    if (c < 15 || c > 17 || table_stride == 0) return false;
    const uint64_t idx = (uint64_t)nwin_for(c) * table_stride;
    uint32_t ib = 24;
    while (((uint64_t)1 << ib) < idx) ib++;
    if (ib > 26) return false;
    const uint32_t fb = 31 - ib > 7 ? 7 : 31 - ib;
    if (((1u << (c - 1)) >> fb) > 512) return false;
    if (ib_out) *ib_out = ib;
    if (fb_out) *fb_out = fb;
    return true;
}
static bool msm_wide_applies(uint32_t c, size_t table_stride) { return msm_wide_shape(c, table_stride, nullptr, nullptr); }
uint32_t msm_auto_window(size_t n, uint32_t override_c) {
    uint32_t lg = 0;
    while (((size_t)1 << (lg + 1)) <= n) lg++;
    // measured on MI355X with whole proofs (tools/single_ab.py, tools/window_exp.py, tools/bench_rows.py): the wide path's 16-bit
    // windows at 2^16 .. 2^20 (16 bucket additions per scalar instead of 20 / 22: k = 19 single proof 13.7 -> 12.3 ms, k = 18
    // 10.3 -> 9.2, k = 17 7.5 -> 6.95 ms and 160 -> 169 proofs/s, k = 16 7.4 -> 7.0); below, lg - 5 bits on the 13-bit plan (k = 15:
    // 7.0 ms against 7.6 / 7.9 with 15 / 16 bits); 2^21 and up (window table indexes beyond 24 bits) 15 bits on the swept sort
    // round 4: 16 bits also at 2^21 (table indexes of 25 bits; was: 15 bits on the swept sort); 17 bits only on request (below)
    int c = lg >= 22 ? 15 : (lg == 21 ? 16 : (lg >= ZK_W17_MIN_LG ? 17 : (lg >= 16 ? 16 : (int)lg - 5)));
    if (override_c) c = (int)override_c;  // zk_ctx_set_option(ZK_OPT_MSM_WINDOW)
    if (c < 9) c = 9;
    if (c > 17) c = 17;
    while (c > 15 && !msm_wide_applies(c, n)) c--;  // 16 / 17-bit digits exist on the wide path only
    // the MSM workspaces are at least 1024 scalars long (get_msm_ws), the wide path needs table stride == workspace length:
    // below that a 15 / 16-bit override would build internal-form tables that msm_run reads on the standard path
    if (c >= 15 && n < 1024) c = 14;
    return (uint32_t)c;
}

// arbitrary bases (generic mode: a bucket set per window, host Horner over the windows): the round-2 rule — lg - 6 from 2^19
// (13 bits at 2^19), 12 at 2^16 .. 2^18, lg - 5 below, at most 14 (15 from 2^21)
uint32_t msm_auto_window_generic(size_t n) {
    uint32_t lg = 0;
    while (((size_t)1 << (lg + 1)) <= n) lg++;
    int c = lg >= 19 ? (int)lg - 6 : (lg >= 16 ? 12 : (int)lg - 5);
    if (c > 14) c = lg >= 21 ? 15 : 14;
    if (c < 9) c = 9;
    return (uint32_t)c;
}

uint32_t msm_num_windows(uint32_t c) { return nwin_for(c); }
size_t msm_ws_max_n(const MsmWorkspace* ws) { return ws->max_n; }
uint32_t msm_ws_max_batch(const MsmWorkspace* ws) { return ws->max_batch; }
uint32_t msm_ws_window(const MsmWorkspace* ws) { return ws->c; }
void msm_ws_set_t1_mode(MsmWorkspace* ws, uint32_t mode) { ws->w_t1_mode = mode; }
bool msm_ws_last_pass_wide(const MsmWorkspace* ws) { return ws->w_last_wide; }

// the kernels, by plan (one translation unit: the parts share the constants and the workspace record above)
#include "msm_legacy.hip.h"  // 13-bit fused plan (n < 2^16), 15-bit swept sort (n >= 2^22), arbitrary bases
#include "msm_wide.hip.h"    // 15 .. 17-bit windows, one bucket set per column: every proof of the reference's configurations
#include "msm_tables.hip.h"  // window tables

// ------------------------------------------------------------------ host ---

#define MSM_TRY(x)                     \
    do {                               \
        hipError_t _e = (x);           \
        if (_e != hipSuccess) {        \
            if (err) *err = _e;        \
            msm_workspace_destroy(ws); \
            return nullptr;            \
        }                              \
    } while (0)

MsmWorkspace* msm_workspace_create(size_t max_n, uint32_t c, hipError_t* err, uint32_t max_batch) {
    if (err) *err = hipSuccess;
    if (c == 0) c = msm_auto_window(max_n);
    if (max_batch == 0) max_batch = 1;
    if (c < 9 || c > 17 || (c >= 16 && !msm_wide_applies(c, max_n)) || max_n == 0 || max_n > ((size_t)1 << 26) || max_batch > MSM_MAX_BATCH) {
        if (err) *err = hipErrorInvalidValue;
        return nullptr;
    }
    MsmWorkspace* ws = new MsmWorkspace();
    memset(ws, 0, sizeof(*ws));
    ws->max_n = max_n;
    ws->max_batch = max_batch;
    ws->c = c;
    ws->nwin = nwin_for(c);
    ws->nb = 1u << (c - 1);
    ws->parts_fixed = 8;
    ws->parts_generic = 1;
    // two shapes share the buffers: generic mode (one column, a bucket set per window) and fixed-base
    // mode (max_batch columns, one bucket set each)
    const size_t slices = ws->nwin > max_batch ? ws->nwin : max_batch;
    const size_t nbt = slices * ws->nb;
    const size_t ent = (size_t)max_batch * max_n * ws->nwin + nbt * (PAD - 1);  // + per-bucket padding
    const size_t nchunks = (max_n + CHUNK - 1) / CHUNK;
    const size_t threads = (ent + SEG0 - 1) / SEG0 + 1;
    size_t part_n = (size_t)ws->nwin * ws->nb * ws->parts_generic;
    if ((size_t)max_batch * ws->nb * ws->parts_fixed > part_n) part_n = (size_t)max_batch * ws->nb * ws->parts_fixed;
    MSM_TRY(hipMalloc(&ws->digits, (size_t)max_batch * max_n * ws->nwin * sizeof(int16_t)));
    MSM_TRY(hipMalloc(&ws->totals, (nbt + 1) * 4));
    MSM_TRY(hipMalloc(&ws->bucket_start, (nbt + 1) * 4));
    {
        const size_t generic_blocks = nchunks * ws->nwin, fixed_blocks = (size_t)max_batch * ((max_n + FCHUNK - 1) / FCHUNK);
        MSM_TRY(hipMalloc(&ws->blockbase, (generic_blocks > fixed_blocks ? generic_blocks : fixed_blocks) * ws->nb * 4));
    }
    MSM_TRY(hipMalloc(&ws->counts, 4 * 4 * (MSM_MAX_BATCH + 1)));
    MSM_TRY(hipMalloc(&ws->entries, ent * sizeof(uint32_t)));
    {
        const size_t nblk_max = (max_n + FCHUNK - 1) / FCHUNK;
        ws->inter_stride = max_n * ws->nwin;
        ws->coarse_stride = (uint32_t)(WHDR + nblk_max * WCB);  // the wide path's layout (512 bins) covers the 13-bit plan's (256)
        MSM_TRY(hipMalloc(&ws->inter, (size_t)max_batch * ws->inter_stride * sizeof(uint32_t)));
        MSM_TRY(hipMalloc(&ws->coarse, (size_t)max_batch * ws->coarse_stride * sizeof(uint32_t)));
        MSM_TRY(hipMalloc(&ws->cursor, (size_t)max_batch * ws->nb * sizeof(uint32_t)));
    }
    MSM_TRY(hipMalloc(&ws->redo, threads * sizeof(uint32_t)));
    ws->slot_elems = threads;
    MSM_TRY(hipMalloc(&ws->slot_pt, threads * sizeof(G1X29S)));
    MSM_TRY(hipMalloc(&ws->partial, (threads / GA + 2) * sizeof(G1X29S)));
    MSM_TRY(hipMalloc(&ws->part, part_n * sizeof(G1X29S)));
    MSM_TRY(hipMalloc(&ws->bit_sum, slices * c * BITSUM_MAX_SPLIT * sizeof(G1X)));
    ws->wide = msm_wide_shape(c, max_n, &ws->w_ib, &ws->w_fb);
    ws->w_clean = false;
    if (ws->wide) {
        ws->w_ent_stride = (max_n * ws->nwin + 63) & ~(size_t)63;
        if (ws->w_ent_stride * max_batch > ent) {  // cannot happen: the dense list is padded to 64 per bucket
            if (err) *err = hipErrorInvalidValue;
            msm_workspace_destroy(ws);
            return nullptr;
        }
        ws->w_lane_stride = (uint32_t)(ws->w_ent_stride / WL) + 64;
        ws->w_slot_stride = ws->w_lane_stride + ws->nb + 64;
        ws->w_part_stride = ((ws->w_lane_stride + 2 * ws->nb) / WCAP_MIN + ws->nb + 2 * (ws->nb >> ws->w_fb) + 127) & ~63u;
        if ((size_t)max_batch * ws->w_slot_stride > threads) {  // cannot happen: sized for 16-entry segments
            if (err) *err = hipErrorInvalidValue;
            msm_workspace_destroy(ws);
            return nullptr;
        }
        ws->w_tot[0] = ws->totals;
        ws->w_cur[0] = ws->cursor;
        MSM_TRY(hipMalloc(&ws->w_tot[1], (size_t)max_batch * ws->nb * 4));
        MSM_TRY(hipMalloc(&ws->w_cur[1], (size_t)max_batch * ws->nb * 4));
        MSM_TRY(hipMalloc(&ws->w_lane_b, (size_t)max_batch * ws->w_lane_stride * 4));
        MSM_TRY(hipMalloc(&ws->w_bstart, (size_t)max_batch * ws->nb * 4));
        MSM_TRY(hipMalloc(&ws->w_pstart, (size_t)max_batch * ws->nb * 4));
        MSM_TRY(hipMalloc(&ws->w_pbucket, (size_t)max_batch * ws->w_part_stride * 4));
        MSM_TRY(hipMalloc(&ws->w_part, (size_t)max_batch * ws->w_part_stride * sizeof(G1X29S)));
        MSM_TRY(hipMalloc(&ws->w_rc, (size_t)max_batch * ((ws->nb >> 8) + 256) * sizeof(G1X29S)));
    }
    return ws;
}

void msm_workspace_destroy(MsmWorkspace* ws) {
    if (!ws) return;
    hipFree(ws->digits);
    hipFree(ws->totals);
    hipFree(ws->bucket_start);
    hipFree(ws->blockbase);
    hipFree(ws->counts);
    hipFree(ws->entries);
    hipFree(ws->inter);
    hipFree(ws->coarse);
    hipFree(ws->cursor);
    hipFree(ws->redo);
    hipFree(ws->slot_pt);
    hipFree(ws->partial);
    hipFree(ws->part);
    hipFree(ws->bit_sum);
    if (ws->wide) {
        hipFree(ws->w_tot[1]);
        hipFree(ws->w_cur[1]);
    }
    hipFree(ws->w_lane_b);
    hipFree(ws->w_bstart);
    hipFree(ws->w_pstart);
    hipFree(ws->w_pbucket);
    hipFree(ws->w_part);
    hipFree(ws->w_rc);
    delete ws;
}

// The wide path's pipeline (see "wide path" above): `batch` columns against the resident basis whose window table — in the
// internal form, msm_build_table(.., internal = true) — is `table`.
static hipError_t msm_run_wide(MsmWorkspace* ws, const Fr* const* scalars_list, uint32_t batch, size_t n, hipStream_t st,
                               G1X* host_window_sums, uint32_t* nwin_out, uint32_t* c_out, hipEvent_t* accum_events,
                               const G1Affine* table, uint32_t table_stride, hipStream_t tail_st, hipEvent_t head_done,
                               bool bases_may_be_identity) {
    const uint32_t c = ws->c, nwin = ws->nwin, nb = ws->nb;
    const uint32_t rows = nb >> 8;
    const uint32_t WCAP = wcap_for(batch);
    uint32_t row_bits = 0;
    while ((1u << row_bits) < rows) row_bits++;
    *nwin_out = batch;
    *c_out = c;
    hipError_t e;
    WideGeo g;
    memset(&g, 0, sizeof(g));
    g.c = c;
    g.nwin = nwin;
    g.nb = nb;
    g.ib = ws->w_ib;
    g.fb = ws->w_fb;
    g.bins = nb >> g.fb;
    for (uint32_t w = 0; w < nwin; w++) {  // the recoding bias: s + K < 2^254 + 2^(nwin c - 1) (1 + 2^-c + ..) < 2^256
        const uint32_t bit = c - 1 + w * c;
        g.K[bit >> 5] |= 1u << (bit & 31);
    }
#ifdef ZK_MSM_POISON  // debug: a slot / part / row-column sum read without having been written by THIS pass shows
    hipMemsetAsync(ws->slot_pt, 0xA5, ws->slot_elems * sizeof(G1X29S), st);
    hipMemsetAsync(ws->w_part, 0xA5, (size_t)ws->max_batch * ws->w_part_stride * sizeof(G1X29S), st);
    hipMemsetAsync(ws->w_rc, 0xA5, (size_t)ws->max_batch * ((ws->nb >> 8) + 256) * sizeof(G1X29S), st);
#endif
    if (!ws->w_clean) {
        // first pass on this workspace (or the previous one did not finish): every later pass finds the counters zeroed by
        // the last tail kernel of the pass before
        const uint32_t all = ws->max_batch * nb;
        uint32_t m = all;
        if (ws->max_batch * WCB > m) m = ws->max_batch * WCB;
        if (4 * (MSM_MAX_BATCH + 1) > m) m = 4 * (MSM_MAX_BATCH + 1);
        hipLaunchKernelGGL(msm_wclear_kernel, dim3((m + 255) / 256), dim3(256), 0, st, ws->w_tot[0], ws->w_cur[0], ws->w_tot[1], ws->w_cur[1],
                           all, ws->counts, ws->coarse, ws->coarse_stride, ws->max_batch);
        ws->w_used_cols[0] = ws->w_used_cols[1] = 0;
    }
    ws->w_clean = false;
    uint32_t* const totals = ws->w_tot[ws->w_par];  // this pass's set of per-bucket counters; H1 zeroes the other one
    uint32_t* const cursor = ws->w_cur[ws->w_par];
    uint32_t* const next_totals = ws->w_tot[ws->w_par ^ 1];
    uint32_t* const next_cursor = ws->w_cur[ws->w_par ^ 1];
    const uint32_t n32 = (uint32_t)n;
    const uint32_t lanes = (uint32_t)(((size_t)n * nwin + WL - 1) / WL);  // accumulation lanes of one column at most
    if (n > 0) {
        const uint32_t nblk = (n32 + FCHUNK - 1) / FCHUNK;
        MsmBatch mb;
        memset(&mb, 0, sizeof(mb));
        for (uint32_t q = 0; q < batch; q++) mb.s[q] = scalars_list[q];
        const dim3 gh(nblk, batch);
        if (c == 17) {
            hipLaunchKernelGGL(msm_whist_kernel<17>, gh, dim3(256), 0, st, mb, n32, g, ws->coarse, ws->coarse_stride, ws->counts);
            hipLaunchKernelGGL(msm_wscatter1_kernel<17>, gh, dim3(256), 0, st, mb, n32, g, table_stride, ws->coarse, ws->coarse_stride,
                               ws->inter, ws->inter_stride, ws->counts, WCAP);
        } else if (c == 16) {
            hipLaunchKernelGGL(msm_whist_kernel<16>, gh, dim3(256), 0, st, mb, n32, g, ws->coarse, ws->coarse_stride, ws->counts);
            hipLaunchKernelGGL(msm_wscatter1_kernel<16>, gh, dim3(256), 0, st, mb, n32, g, table_stride, ws->coarse, ws->coarse_stride,
                               ws->inter, ws->inter_stride, ws->counts, WCAP);
        } else {
            hipLaunchKernelGGL(msm_whist_kernel<15>, gh, dim3(256), 0, st, mb, n32, g, ws->coarse, ws->coarse_stride, ws->counts);
            hipLaunchKernelGGL(msm_wscatter1_kernel<15>, gh, dim3(256), 0, st, mb, n32, g, table_stride, ws->coarse, ws->coarse_stride,
                               ws->inter, ws->inter_stride, ws->counts, WCAP);
        }
        const uint32_t max_chunks = (uint32_t)(((uint64_t)n32 * nwin + SUB - 1) / SUB) + g.bins;
        hipLaunchKernelGGL(msm_wfinehist_kernel, dim3(max_chunks + 1, batch), dim3(256), 0, st, ws->inter, ws->inter_stride, ws->coarse,
                           ws->coarse_stride, g, totals, next_totals, next_cursor, ws->w_used_cols[ws->w_par ^ 1]);
        hipLaunchKernelGGL(msm_wscatter2_kernel, dim3(max_chunks + 1, batch), dim3(256), 0, st, ws->inter, ws->inter_stride, ws->coarse,
                           ws->coarse_stride, g, totals, ws->w_bstart, cursor, ws->entries, ws->w_ent_stride, ws->w_lane_b, ws->w_lane_stride,
                           ws->w_pstart, ws->w_pbucket, ws->w_part_stride, WCAP);
        if (accum_events) hipEventRecord(accum_events[0], st);
        if (bases_may_be_identity) {
            hipLaunchKernelGGL(msm_wacc_kernel, dim3((lanes + 63) / 64, batch), dim3(64), 0, st, ws->entries, ws->w_ent_stride, table, ws->counts,
                               ws->w_lane_b, ws->w_lane_stride, ws->w_bstart, nb, ws->slot_pt, ws->w_slot_stride, g.ib);
        } else {
            hipLaunchKernelGGL(msm_wacc_fast_kernel, dim3((lanes + 63) / 64, batch), dim3(64), 0, st, ws->entries, ws->w_ent_stride, table,
                               ws->counts, ws->w_lane_b, ws->w_lane_stride, ws->w_bstart, nb, ws->slot_pt, ws->w_slot_stride, ws->redo, g.ib);
        }
        if (accum_events) hipEventRecord(accum_events[1], st);
    }
    hipStream_t ts = st;
    if (tail_st && tail_st != st) {
        if ((e = hipEventRecord(head_done, st)) != hipSuccess) return e;
        if ((e = hipStreamWaitEvent(tail_st, head_done, 0)) != hipSuccess) return e;
        ts = tail_st;
    }
    if (accum_events && accum_events[2]) hipEventRecord(accum_events[2], ts);  // the reduction tail (T1 .. T3) alone
    if (n > 0) {
        // parts of one column at most: every bin's region is its slots / WCAP + a part per bucket + slack (msm_wscatter1_kernel)
        const uint32_t max_parts = (lanes + 2 * nb) / WCAP + nb + 2 * g.bins + 64;
        // T1 per part (default) or per bucket (ZK_OPT_MSM_T1 = 1: fewer instructions, a longer chain — measured: no gain, msm_wide.hip.h)
        const bool per_bucket = ws->w_t1_mode == 1;
        uint32_t lmin = 0;
        if (per_bucket) {
            lmin = ZK_T1B_LMAX;
            hipLaunchKernelGGL(msm_wbucket_kernel, dim3(nb / 64, batch), dim3(64), 0, ts, ws->slot_pt, ws->w_slot_stride, totals, ws->w_bstart,
                               ws->w_pstart, ws->w_part_stride, nb, ws->w_part, WCAP, lmin);
        }
        hipLaunchKernelGGL(msm_wparts_kernel, dim3((max_parts + 63) / 64, batch), dim3(64), 0, ts, ws->slot_pt, ws->w_slot_stride, totals,
                           ws->w_bstart, ws->w_pstart, ws->w_pbucket, ws->w_part_stride, nb, ws->counts, ws->w_part, WCAP, lmin);
    }
    hipLaunchKernelGGL(msm_wrowcol_kernel, dim3(rows + 256, batch), dim3(64), 0, ts, ws->w_part, ws->w_part_stride, totals, ws->w_bstart,
                       ws->w_pstart, nb, ws->w_rc, WCAP);
    // T3 writes its sums (and the redo count) STRAIGHT into the caller's pinned host buffer: 17 x 128 bytes per column over the
    // bus instead of a copy kernel at the end of every pass's chain (host_window_sums must be device-visible pinned memory: the
    // engine's lanes allocate it with hipHostMalloc)
    G1X* out_dev = nullptr;
    if ((e = hipHostGetDevicePointer((void**)&out_dev, host_window_sums, 0)) != hipSuccess) return e;
    hipLaunchKernelGGL(msm_wbits_kernel, dim3(9 + row_bits, batch), dim3(64), 0, ts, ws->w_rc, nb, out_dev, ws->counts);
    if (accum_events && accum_events[3]) hipEventRecord(accum_events[3], ts);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    ws->w_clean = true;
    ws->w_redo_valid = n > 0 && !bases_may_be_identity;  // (the checked loop lists nothing; an empty pass never reset counts[1])
    if (n > 0) {  // (an empty pass neither counts nor zeroes anything)
        ws->w_used_cols[ws->w_par] = batch;
        ws->w_used_cols[ws->w_par ^ 1] = 0;
        ws->w_par ^= 1;  // the next pass counts in the set this one has zeroed
    }
    return hipSuccess;
}

// The unchecked accumulation of the wide path lists the lanes whose sums show an exceptional step; the count travels to the host
// with the pass's sums (msm_wbits_kernel).  Non-zero — a degenerate basis only — and the host calls this: the listed lanes again
// with the checked loop, then the tail again, on `st`; the pass's workspace is intact until the lane's next pass.
uint32_t msm_wide_redo_count(const MsmWorkspace* ws, const G1X* host_window_sums, uint32_t batch) {
    if (!ws->wide || !ws->w_redo_valid) return 0;
    uint32_t v;
    memcpy(&v, host_window_sums + (size_t)batch * WIDE_SUMS, 4);
    return v;
}

hipError_t msm_wide_redo(MsmWorkspace* ws, uint32_t batch, size_t n, hipStream_t st, G1X* host_window_sums, const G1Affine* table) {
    if (!ws->wide || n == 0 || batch == 0 || batch > ws->max_batch) return hipErrorInvalidValue;
    const uint32_t nb = ws->nb, rows = nb >> 8, nwin = ws->nwin, WCAP = wcap_for(batch);
    uint32_t row_bits = 0;
    while ((1u << row_bits) < rows) row_bits++;
    const uint32_t lanes = (uint32_t)(((size_t)n * nwin + WL - 1) / WL);
    const uint32_t bins = nb >> ws->w_fb;
    uint32_t* const totals = ws->w_tot[ws->w_par ^ 1];  // the set the pass counted in (the parity has moved on)
    hipLaunchKernelGGL(msm_wacc_redo_kernel, dim3(256), dim3(64), 0, st, ws->entries, ws->w_ent_stride, table, ws->counts, ws->w_lane_b,
                       ws->w_lane_stride, ws->w_bstart, nb, ws->slot_pt, ws->w_slot_stride, ws->redo, ws->w_ib);
    hipError_t e = hipMemsetAsync(ws->counts + 1, 0, 4, st);  // the second tail reports none
    if (e != hipSuccess) return e;
    const uint32_t max_parts = (lanes + 2 * nb) / WCAP + nb + 2 * bins + 64;
    hipLaunchKernelGGL(msm_wparts_kernel, dim3((max_parts + 63) / 64, batch), dim3(64), 0, st, ws->slot_pt, ws->w_slot_stride, totals,
                       ws->w_bstart, ws->w_pstart, ws->w_pbucket, ws->w_part_stride, nb, ws->counts, ws->w_part, WCAP, 0u);
    hipLaunchKernelGGL(msm_wrowcol_kernel, dim3(rows + 256, batch), dim3(64), 0, st, ws->w_part, ws->w_part_stride, totals, ws->w_bstart,
                       ws->w_pstart, nb, ws->w_rc, WCAP);
    G1X* out_dev = nullptr;
    if ((e = hipHostGetDevicePointer((void**)&out_dev, host_window_sums, 0)) != hipSuccess) return e;
    hipLaunchKernelGGL(msm_wbits_kernel, dim3(9 + row_bits, batch), dim3(64), 0, st, ws->w_rc, nb, out_dev, ws->counts);
    return hipGetLastError();
}

// `table` != nullptr selects the fixed-base mode: table[w * table_stride + i] = 2^(c w) P_i.
// The "head" (recode .. accumulate) runs on `st`; the latency-bound "tail" (gather, bit sums,
// D2H) runs on `tail_st` after `head_done`, so that the caller can put the next MSM's head (or
// any other kernels) on `st` right away.  tail_st == st gives the plain sequential order.
hipError_t msm_run(MsmWorkspace* ws, const Fr* const* scalars_list, uint32_t batch, const G1Affine* bases, size_t n,
                   hipStream_t st, G1X* host_window_sums, uint32_t* nwin_out, uint32_t* c_out, hipEvent_t* accum_events,
                   const G1Affine* table, uint32_t table_stride, hipStream_t tail_st, hipEvent_t head_done,
                   bool bases_may_be_identity) {
    if (n > ws->max_n || batch == 0 || batch > ws->max_batch) return hipErrorInvalidValue;
    const uint32_t c = ws->c, nwin = ws->nwin, nb = ws->nb;
    const bool fixed = table != nullptr;
    ws->w_redo_valid = false;
    ws->w_last_wide = fixed && ws->wide && table_stride == ws->max_n;
    if (fixed && ws->wide && table_stride == ws->max_n)
        return msm_run_wide(ws, scalars_list, batch, n, st, host_window_sums, nwin_out, c_out, accum_events, table, table_stride, tail_st,
                            head_done, bases_may_be_identity);
    if (c > 15) return hipErrorInvalidValue;  // 16 / 17-bit digits exist on the wide path only
    ws->w_clean = false;                       // this plan shares totals / cursors / counts with the wide path and leaves them used
    const bool fused = fixed && nb <= SORT_LDS_BUCKETS;  // one-kernel digits + histogram; 15-bit windows take the swept sort
    if (!fused && batch != 1) return hipErrorInvalidValue;  // columns are batched on the fused fixed-base path only
    const Fr* scalars = scalars_list[0];
    const uint32_t slices = fixed ? batch : nwin;
    const uint32_t nbt = slices * nb;
    const uint32_t parts = fixed ? ws->parts_fixed : ws->parts_generic;
    *nwin_out = slices;
    *c_out = c;
    hipError_t e;
    {
        // the bucket parts need no reset: the bit sums read only the parts the gather wrote (msm_bitsum_kernel used_parts);
        // an empty MSM runs no gather, so its parts are set to the identity here
#ifdef ZK_MSM_POISON
        hipMemsetAsync(ws->part, 0xA5, (size_t)nbt * parts * sizeof(G1X29S), st);  // debug: any part read without being written shows
#endif
        const uint32_t clear_parts = n > 0 ? 0u : nbt * parts;
        uint32_t m = nbt + 1;
        if (clear_parts > m) m = clear_parts;
        if (batch * CBINS_MAX > m) m = batch * CBINS_MAX;
        const bool sort2 = sort2_applies(fused, n, nb, nwin, table_stride);
        hipLaunchKernelGGL(msm_clear_kernel, dim3((m + 255) / 256), dim3(256), 0, st, ws->part, clear_parts, ws->totals, nbt + 1,
                           ws->counts, ws->cursor, sort2 ? nbt : 0u, ws->coarse, ws->coarse_stride, sort2 ? batch : 0u);
    }
    if (n > 0) {
        const uint32_t n32 = (uint32_t)n;
        const uint32_t stride = (uint32_t)ws->max_n;
        const uint32_t nchunks = (n32 + CHUNK - 1) / CHUNK;
        const bool sort2 = sort2_applies(fused, n, nb, nwin, table_stride);
        if (sort2) {
            // two-level counting sort with coalesced stores (see msm_scatter1_kernel)
            const uint32_t nblk = (n32 + FCHUNK - 1) / FCHUNK;
            MsmBatch mb;
            memset(&mb, 0, sizeof(mb));
            for (uint32_t q = 0; q < batch; q++) mb.s[q] = scalars_list[q];
            hipLaunchKernelGGL(msm_recode_hist2_kernel, dim3(nblk, batch), dim3(256), (nb + 256 * 9) * 4, st, mb, n32, stride, c, nwin, nb,
                               ws->digits, ws->totals, ws->coarse, ws->coarse_stride);
            hipLaunchKernelGGL(msm_scan_kernel, dim3(1), dim3(1024), 0, st, ws->totals, ws->bucket_start, nbt, ws->counts);
            hipLaunchKernelGGL(msm_scan_coarse_kernel, dim3(batch), dim3(CBINS_MAX), 0, st, ws->totals, nb, ws->coarse, ws->coarse_stride, nb >> 6, 6u);
            hipLaunchKernelGGL(msm_scatter1_kernel, dim3(nblk, batch), dim3(256), 0, st, ws->digits, n32, stride, nwin, nb, table_stride,
                               ws->coarse, ws->coarse_stride, ws->inter, ws->inter_stride, 6u);
            const uint32_t max_chunks = (uint32_t)(((uint64_t)n32 * nwin + SUB - 1) / SUB) + (nb >> 6);
            hipLaunchKernelGGL(msm_scatter2_kernel, dim3(max_chunks, batch), dim3(256), 0, st, ws->inter, ws->inter_stride, ws->coarse,
                               ws->coarse_stride, nb, ws->totals, ws->bucket_start, ws->cursor, ws->entries, 6u, PAD, (size_t)0, (const uint8_t*)nullptr);
        } else if (fused) {
            const uint32_t nblk = (n32 + FCHUNK - 1) / FCHUNK;
            MsmBatch mb;
            memset(&mb, 0, sizeof(mb));
            for (uint32_t q = 0; q < batch; q++) mb.s[q] = scalars_list[q];
            hipLaunchKernelGGL(msm_recode_hist_kernel, dim3(nblk, batch), dim3(256), (nb + 256 * 9) * 4, st, mb, n32, stride, c,
                               nwin, nb, ws->digits, ws->totals, ws->blockbase);
            hipLaunchKernelGGL(msm_scan_kernel, dim3(1), dim3(1024), 0, st, ws->totals, ws->bucket_start, nbt, ws->counts);
            // Long columns: one scatter launch per column — a column's 4-byte stores land in its own ~40 MB of the
            // entry list and combine in the cache; all columns at once thrash it (0.8 ms instead of 4 x 0.1 ms at
            // four columns of 2^19).  Short columns: one launch for all (launch-bound otherwise).
            const uint32_t per_launch = n32 >= (1u << 18) ? 1 : batch;
            for (uint32_t q = 0; q < batch; q += per_launch)
                hipLaunchKernelGGL(msm_scatter_fixed_kernel, dim3(nblk, per_launch), dim3(SCAT_T), nb * 4, st,
                                   ws->digits + (size_t)q * nwin * stride, n32, stride, nwin, nb, table_stride,
                                   ws->totals + (size_t)q * nb, ws->bucket_start + (size_t)q * nb,
                                   ws->blockbase + (size_t)q * nblk * nb, ws->entries);
        } else {
            hipLaunchKernelGGL(msm_recode_kernel, dim3((n32 + 255) / 256), dim3(256), 0, st, scalars, n32, stride, c, nwin,
                               ws->digits);
            hipLaunchKernelGGL(msm_sort_kernel<false>, dim3(nchunks * nwin), dim3(256), (nb < SORT_LDS_BUCKETS ? nb : SORT_LDS_BUCKETS) * 4, st, ws->digits, n32, stride,
                               nchunks, nb, fixed ? 1u : 0u, table_stride, ws->totals, ws->bucket_start, ws->blockbase,
                               (uint32_t*)nullptr);
            hipLaunchKernelGGL(msm_scan_kernel, dim3(1), dim3(1024), 0, st, ws->totals, ws->bucket_start, nbt, ws->counts);
            hipLaunchKernelGGL(msm_sort_kernel<true>, dim3(nchunks * nwin), dim3(256), (nb < SORT_LDS_BUCKETS ? nb : SORT_LDS_BUCKETS) * 4, st, ws->digits, n32, stride,
                               nchunks, nb, fixed ? 1u : 0u, table_stride, ws->totals, ws->bucket_start, ws->blockbase,
                               ws->entries);
            hipLaunchKernelGGL(msm_pad_kernel, dim3((nbt + 255) / 256), dim3(256), 0, st, ws->totals, ws->bucket_start, nbt,
                               ws->entries);
        }
        const size_t worst = (size_t)(fixed ? batch : 1) * n * nwin + (size_t)nbt * (PAD - 1);  // worst-case padded entry count
        const size_t threads = (worst + SEG0 - 1) / SEG0;
        if (accum_events) hipEventRecord(accum_events[0], st);
        if (bases_may_be_identity) {
            hipLaunchKernelGGL(msm_accumulate_kernel, dim3((uint32_t)((threads + 63) / 64)), dim3(64), 0, st, ws->entries,
                               fixed ? table : bases, ws->counts, ws->slot_pt);
        } else {
            hipLaunchKernelGGL(msm_accumulate_fast_kernel, dim3((uint32_t)((threads + 63) / 64)), dim3(64), 0, st, ws->entries,
                               fixed ? table : bases, ws->counts, ws->redo, ws->slot_pt);
        }
        if (accum_events) hipEventRecord(accum_events[1], st);  // the dominant kernel alone (bench.py's roofline)
        if (!bases_may_be_identity)
            hipLaunchKernelGGL(msm_accumulate_redo_kernel, dim3(1024), dim3(64), 0, st, ws->entries, fixed ? table : bases,
                               ws->counts, ws->redo, ws->slot_pt);
    }
    hipStream_t ts = st;
    if (tail_st && tail_st != st) {
        if ((e = hipEventRecord(head_done, st)) != hipSuccess) return e;
        if ((e = hipStreamWaitEvent(tail_st, head_done, 0)) != hipSuccess) return e;
        ts = tail_st;
    }
    if (n > 0) {
        const size_t worst = (size_t)(fixed ? batch : 1) * n * nwin + (size_t)nbt * (PAD - 1);
        const size_t lanes1 = (worst + PAD - 1) / PAD;
        hipLaunchKernelGGL(msm_gather1_kernel, dim3((uint32_t)((lanes1 + 63) / 64)), dim3(64), 0, ts, ws->slot_pt, ws->counts,
                           ws->partial);
        const uint32_t ngroups = nbt * parts;
        if (fixed)
            hipLaunchKernelGGL(msm_gather_kernel<GLANES_FIXED>, dim3((ngroups * GLANES_FIXED + 255) / 256), dim3(256), 0, ts, ws->bucket_start,
                               ws->partial, parts, ngroups, ws->part);
        else
            hipLaunchKernelGGL(msm_gather_kernel<4>, dim3((ngroups * 4 + 255) / 256), dim3(256), 0, ts, ws->bucket_start,
                               ws->partial, parts, ngroups, ws->part);
    }
    const uint32_t bt = bitsum_threads(nb), bs = bitsum_split(nb);
    if (bt == 512)
        hipLaunchKernelGGL(msm_bitsum_kernel<512>, dim3(slices * c * bs), dim3(512), 0, ts, ws->part, parts, nb, c, bs, ws->bucket_start,
                           ws->bit_sum);
    else if (bt == 256)
        hipLaunchKernelGGL(msm_bitsum_kernel<256>, dim3(slices * c * bs), dim3(256), 0, ts, ws->part, parts, nb, c, bs, ws->bucket_start,
                           ws->bit_sum);
    else if (bt == 128)
        hipLaunchKernelGGL(msm_bitsum_kernel<128>, dim3(slices * c * bs), dim3(128), 0, ts, ws->part, parts, nb, c, bs, ws->bucket_start,
                           ws->bit_sum);
    else
        hipLaunchKernelGGL(msm_bitsum_kernel<64>, dim3(slices * c * bs), dim3(64), 0, ts, ws->part, parts, nb, c, bs, ws->bucket_start,
                           ws->bit_sum);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    return hipMemcpyAsync(host_window_sums, ws->bit_sum, (size_t)slices * c * bs * sizeof(G1X),
                          hipMemcpyDeviceToHost, ts);
}

// bit_sums[(w * c + t) * split + q]: partials of G_{w,t}, split = bitsum_split(2^(c-1));
// result = sum_w 2^(c w) sum_t 2^t G_{w,t}  (Horner over all bit positions)
uint32_t msm_sums_per_result(uint32_t c) { return c * bitsum_split(1u << (c - 1)); }

uint32_t msm_ws_sums_per_result(const MsmWorkspace* ws) { return ws->wide ? WIDE_SUMS : msm_sums_per_result(ws->c); }

// fixed-base mode: one column's result from its bit sums
G1Jac msm_ws_finish_fixed(const MsmWorkspace* ws, const G1X* sums) {
    if (!ws->wide) return msm_finish_host(sums, 1, ws->c);
    // sum_{t < 9} 2^t S_t (columns of the bucket matrix, weight l + 1) + sum_u 2^(8 + u) S_{9 + u} (rows, weight 256 h)
    uint32_t row_bits = 0;
    while ((1u << row_bits) < (ws->nb >> 8)) row_bits++;
    G1X acc = G1X::identity();
    for (int pos = (int)(8 + row_bits) - 1; pos >= 0; pos--) {
        if (!acc.is_identity()) acc = g1x_dbl(acc);
        if (pos >= 8) g1x_add(acc, sums[9 + (pos - 8)]);
        if (pos <= 8) g1x_add(acc, sums[pos]);
    }
    return g1x_to_jac(acc);
}

G1Jac msm_finish_host(const G1X* bit_sums, uint32_t nwin, uint32_t c) {
    const uint32_t split = bitsum_split(1u << (c - 1));
    G1X acc = G1X::identity();
    for (int q = (int)(nwin * c) - 1; q >= 0; q--) {
        if (!acc.is_identity()) acc = g1x_dbl(acc);
        for (uint32_t k = 0; k < split; k++) g1x_add(acc, bit_sums[(size_t)q * split + k]);
    }
    return g1x_to_jac(acc);
}

}  // namespace zk
