// msm.hip — Pippenger bucket multi-scalar multiplication over BN254 G1 for gfx950.
//
// Device replacement for halo2_proofs `arithmetic::best_multiexp`, reached only
// through `ParamsKZG::{commit, commit_lagrange}` (reference
// call sites halo2-circuits/src/ecc/ecdsa_p256.rs:366-373,416-423,555-562).
// Any correct algorithm yields the same group element, so the result is
// bit-identical to the reference's after affine normalisation.
//
// Two modes share one pipeline:
//   generic   bases are arbitrary (the fine-grained `zk_msm_bn254` seam): W windows of
//             c bits, W x 2^(c-1) buckets, host does the final W-window Horner;
//   fixed     bases are the resident SRS (`zk_commit`): a table of window multiples
//             2^(c w) P_i is precomputed once per basis, so every window's digits fall
//             into ONE set of 2^(c-1) buckets — no per-window reduction, no Horner.
//
// Pipeline (one stream, no host round trip until the window sums):
//   1. msm_recode        signed c-bit window recoding of every scalar -> digits[W][n]
//   2. msm_sort<COUNT>   per (scalar chunk, window) workgroup: bucket histogram in LDS,
//                        one returning atomicAdd per non-empty bucket reserves the
//                        workgroup's range inside the bucket
//   3. msm_scan          exclusive scan of bucket totals -> bucket starts
//   4. msm_sort<SCATTER> same workgroups: LDS cursors = bucket start + reserved base,
//                        counting-sort scatter of (bucket, +-base index) entries
//   5. msm_accumulate    bucket ranges are padded to multiples of 16 entries; every lane sums one
//                        aligned 16-entry segment with XYZZ mixed adds (8M + 2S) -> one partial
//                        sum ("slot") per lane, all lanes of a launch do the same amount of work
//   6. msm_gather1/2     two-level sum of each bucket's slots: dense lanes add 4 slots serially, then
//                        16-lane groups finish each bucket with a shuffle tree (tail stream)
//   7. msm_bitsum        sum_j j B_j = sum_t 2^t G_t, G_t = sum of buckets with bit t of j set:
//                        c tree reductions per bucket set; the short Horner is done on the host
// Load balance does not depend on the scalar distribution: witness columns are
// dominated by zeros / small values (hot low buckets), and a segment is a fixed number
// of entries whatever bucket they fall in.
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
    uint32_t w_t1_mode;         // T1 of the wide path: 0 auto (per bucket when the tail runs on the head's stream), 1 per bucket, 2 per part
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

// ---------------------------------------------------------------- recode ---

__global__ __launch_bounds__(256) void msm_recode_kernel(const Fr* __restrict__ scalars, uint32_t n, uint32_t stride,
                                                         uint32_t c, uint32_t nwin, int16_t* __restrict__ digits) {
    __shared__ uint32_t limbs[256][9];
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const Fr s = fe_from_mont(fe_load(scalars + i));
    uint32_t* L = limbs[threadIdx.x];
#pragma unroll
    for (int k = 0; k < 8; k++) L[k] = s.v[k];
    L[8] = 0;
    const uint32_t half = 1u << (c - 1);
    const uint32_t mask = (1u << c) - 1;
    uint32_t carry = 0;
    for (uint32_t w = 0; w < nwin; w++) {
        const uint32_t bit = w * c, word = bit >> 5, off = bit & 31;
        uint32_t raw = 0;
        if (word < 8) {
            const uint64_t two = (uint64_t)L[word] | ((uint64_t)L[word + 1] << 32);
            raw = (uint32_t)(two >> off) & mask;
        }
        raw += carry;
        int32_t d;
        if (raw > half) {
            d = (int32_t)raw - (int32_t)(1u << c);
            carry = 1;
        } else {
            d = (int32_t)raw;
            carry = 0;
        }
        digits[(size_t)w * stride + i] = (int16_t)d;
    }
}

// ------------------------------------------------------- histogram / scatter ---
// grid.x = nchunks * nwin; blk = w * nchunks + chunk.  slice = fixed ? 0 : w.
template <bool SCATTER>
__global__ __launch_bounds__(256) void msm_sort_kernel(const int16_t* __restrict__ digits, uint32_t n, uint32_t stride,
                                                       uint32_t nchunks, uint32_t nb, uint32_t fixed,
                                                       uint32_t table_stride, uint32_t* __restrict__ totals,
                                                       const uint32_t* __restrict__ bucket_start,
                                                       uint32_t* __restrict__ blockbase, uint32_t* __restrict__ entries) {
    extern __shared__ uint32_t lds[];  // min(nb, SORT_LDS_BUCKETS) counters / cursors
    const uint32_t blk = blockIdx.x;
    const uint32_t w = blk / nchunks, chunk = blk - w * nchunks;
    const uint32_t slice = fixed ? 0 : w;
    const uint32_t lo = chunk * CHUNK, hi = min(n, lo + CHUNK);
    const int16_t* dg = digits + (size_t)w * stride;
    // bucket sets beyond the LDS budget (c = 15) are handled in several sweeps over the chunk's digits
    const uint32_t span = min(nb, SORT_LDS_BUCKETS);
    for (uint32_t b0 = 0; b0 < nb; b0 += span) {
        if (!SCATTER) {
            for (uint32_t b = threadIdx.x; b < span; b += 256) lds[b] = 0;
        } else {
            for (uint32_t b = threadIdx.x; b < span; b += 256)
                lds[b] = bucket_start[slice * nb + b0 + b] + blockbase[(size_t)blk * nb + b0 + b];
        }
        __syncthreads();
        for (uint32_t i = lo + threadIdx.x; i < hi; i += 256) {
            const int32_t d = dg[i];
            if (d == 0) continue;
            const uint32_t mag = d < 0 ? (uint32_t)(-d) : (uint32_t)d;
            const uint32_t rel = mag - 1 - b0;
            if (rel >= span) continue;
            const uint32_t pos = atomicAdd(&lds[rel], 1u);
            if (SCATTER) {
                const uint32_t idx = (fixed ? w * table_stride : 0) + i;
                entries[pos] = idx | (d < 0 ? SIGN_BIT : 0);
            }
        }
        __syncthreads();
        if (!SCATTER) {
            for (uint32_t b = threadIdx.x; b < span; b += 256) {
                const uint32_t cnt = lds[b];
                blockbase[(size_t)blk * nb + b0 + b] = cnt ? atomicAdd(&totals[slice * nb + b0 + b], cnt) : 0;
            }
            __syncthreads();
        }
    }
}

// ---- fixed-base mode: every window feeds ONE bucket set, so a workgroup owns a chunk of scalars with
// ALL their windows: digit extraction and the LDS histogram are one kernel, and the scatter re-reads the
// digits it wrote (5 launches per MSM head instead of 10).
#ifndef ZK_FCHUNK
#define ZK_FCHUNK 1024
#endif
static constexpr uint32_t FCHUNK = ZK_FCHUNK;  // scalars per workgroup (x nwin entries)

__device__ __forceinline__ uint32_t msm_digits_of(const uint32_t* L, uint32_t c, uint32_t nwin, uint32_t i, uint32_t stride,
                                                  int16_t* __restrict__ digits, uint32_t* hist) {
    const uint32_t half = 1u << (c - 1);
    const uint32_t mask = (1u << c) - 1;
    uint32_t carry = 0;
    for (uint32_t w = 0; w < nwin; w++) {
        const uint32_t bit = w * c, word = bit >> 5, off = bit & 31;
        uint32_t raw = 0;
        if (word < 8) {
            const uint64_t two = (uint64_t)L[word] | ((uint64_t)L[word + 1] << 32);
            raw = (uint32_t)(two >> off) & mask;
        }
        raw += carry;
        int32_t d;
        if (raw > half) {
            d = (int32_t)raw - (int32_t)(1u << c);
            carry = 1;
        } else {
            d = (int32_t)raw;
            carry = 0;
        }
        digits[(size_t)w * stride + i] = (int16_t)d;
        if (d != 0) atomicAdd(&hist[(d < 0 ? -d : d) - 1], 1u);
    }
    return carry;
}

// Batched form: blockIdx.y is the column (one scalar vector each, same bases): every column has its own
// bucket set [col * nb, (col + 1) * nb) and its own digit planes, so that ONE accumulate launch serves
// all columns of a batch.
__global__ __launch_bounds__(256) void msm_recode_hist_kernel(MsmBatch batch, uint32_t n, uint32_t stride, uint32_t c,
                                                              uint32_t nwin, uint32_t nb, int16_t* __restrict__ digits_all,
                                                              uint32_t* __restrict__ totals_all,
                                                              uint32_t* __restrict__ blockbase_all) {
    extern __shared__ uint32_t lds[];  // nb counters, then 256 x 9 limbs
    const uint32_t col = blockIdx.y;
    const Fr* __restrict__ scalars = batch.s[col];
    int16_t* __restrict__ digits = digits_all + (size_t)col * nwin * stride;
    uint32_t* __restrict__ totals = totals_all + (size_t)col * nb;
    uint32_t* __restrict__ blockbase = blockbase_all + (size_t)col * gridDim.x * nb;
    uint32_t* hist = lds;
    uint32_t* L = lds + nb + threadIdx.x * 9;
    for (uint32_t b = threadIdx.x; b < nb; b += 256) hist[b] = 0;
    __syncthreads();
    const uint32_t lo = blockIdx.x * FCHUNK, hi = min(n, lo + FCHUNK);
    for (uint32_t i = lo + threadIdx.x; i < hi; i += 256) {
        const Fr s = fe_from_mont(fe_load(scalars + i));
#pragma unroll
        for (int k = 0; k < 8; k++) L[k] = s.v[k];
        L[8] = 0;
        msm_digits_of(L, c, nwin, i, stride, digits, hist);
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nb; b += 256) {
        const uint32_t cnt = hist[b];
        blockbase[(size_t)blockIdx.x * nb + b] = cnt ? atomicAdd(&totals[b], cnt) : 0;
    }
}

// ---- two-level counting sort (ZK_SORT2).  The one-pass scatter below writes every entry to a random place of a 40 MB
// list: 10.5 M uncoalesced 4-byte stores, 128 us per 2^19 column against 24 us for the same kernel with coalesced stores
// (tools/scat_exp.sh).  Here the entries are first grouped by coarse bin (bucket / 64) and then, inside a bin, by bucket; in
// both levels a workgroup sorts 4096 entries in LDS by a 6-bit key and writes them out in runs (~64 entries = 256 B per key),
// so that consecutive lanes store to consecutive addresses.  The 6-bit fine key rides in bits 24..29 of the entry between
// the two levels (an entry is sign << 31 | window * n + i, which needs 24 bits up to 20 windows x 2^19).
#ifndef ZK_SORT2
#define ZK_SORT2 1
#endif
static constexpr uint32_t CBINS_MAX = 256;   // coarse bins: buckets / 64 (64 at 13-bit windows, 128 at 14); buckets / 128 on the wide path (256 at 16)
#ifndef ZK_SORT_SUB
#define ZK_SORT_SUB 4096
#endif
#ifndef ZK_SORT2_MIN_N
#define ZK_SORT2_MIN_N (1u << 18)
#endif
// The two extra launches and the 4096-entry sub-rounds only pay for long columns: single proofs of the k <= 16 rows of
// bench_ecdsa.config are 2-6 % slower with it, k = 17 1-2 %, k >= 18 equal, and batches of k = 19 proofs 4 % faster.
static bool sort2_applies(bool fused, size_t n, uint32_t nb, uint32_t nwin, size_t table_stride) {
    return ZK_SORT2 && fused && n >= ZK_SORT2_MIN_N && nb >= 64 && (nb >> 6) <= CBINS_MAX &&
           (uint64_t)nwin * table_stride <= (1u << 24);
}

static constexpr uint32_t SUB = ZK_SORT_SUB;  // entries sorted in LDS at a time
static constexpr uint32_t COARSE_WORDS = 5 * (CBINS_MAX + 1);  // per column, CBINS_MAX + 1 words each: bin starts, chunk prefix, append cursors, (unused), wide path: the bins' part regions

// digits + fine histogram: the global bucket totals (the workgroup's counts are added with one atomic per non-empty bucket)
__global__ __launch_bounds__(256) void msm_recode_hist2_kernel(MsmBatch batch, uint32_t n, uint32_t stride, uint32_t c, uint32_t nwin,
                                                               uint32_t nb, int16_t* __restrict__ digits_all,
                                                               uint32_t* __restrict__ totals_all, uint32_t* __restrict__ coarse_all,
                                                               uint32_t coarse_stride) {
    extern __shared__ uint32_t lds[];  // nb counters, then 256 x 9 limbs
    const uint32_t col = blockIdx.y;
    const Fr* __restrict__ scalars = batch.s[col];
    int16_t* __restrict__ digits = digits_all + (size_t)col * nwin * stride;
    uint32_t* __restrict__ totals = totals_all + (size_t)col * nb;
    uint32_t* __restrict__ chdr = coarse_all + (size_t)col * coarse_stride;
    uint32_t* hist = lds;
    uint32_t* L = lds + nb + threadIdx.x * 9;
    for (uint32_t b = threadIdx.x; b < nb; b += 256) hist[b] = 0;
    __syncthreads();
    const uint32_t lo = blockIdx.x * FCHUNK, hi = min(n, lo + FCHUNK);
    for (uint32_t i = lo + threadIdx.x; i < hi; i += 256) {
        const Fr s = fe_from_mont(fe_load(scalars + i));
#pragma unroll
        for (int k = 0; k < 8; k++) L[k] = s.v[k];
        L[8] = 0;
        msm_digits_of(L, c, nwin, i, stride, digits, hist);
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nb; b += 256) {
        const uint32_t cnt = hist[b];
        if (cnt) atomicAdd(&totals[b], cnt);
    }
    // the workgroup's range inside every coarse bin of `inter`: one returning atomic per bin on the append cursors
    const uint32_t bins = nb >> 6;
    if (threadIdx.x < bins) {
        uint32_t sum = 0;
        for (uint32_t q = 0; q < 64; q++) sum += hist[threadIdx.x * 64 + ((q + threadIdx.x) & 63)];  // staggered: no bank conflict
        chdr[COARSE_WORDS + (size_t)blockIdx.x * CBINS_MAX + threadIdx.x] = sum ? atomicAdd(&chdr[2 * (CBINS_MAX + 1) + threadIdx.x], sum) : 0;
    }
}

// per column: coarse-bin totals (sums of 64 bucket totals), their exclusive scan (the bins' places in `inter`), the chunk
// prefix of the second level (ceil(total / SUB) chunks per bin), and the first level's append cursors (zero)
__global__ __launch_bounds__(CBINS_MAX) void msm_scan_coarse_kernel(const uint32_t* __restrict__ totals_all, uint32_t nb,
                                                                    uint32_t* __restrict__ coarse_all, uint32_t coarse_stride, uint32_t bins,
                                                                    uint32_t fb) {
    __shared__ uint32_t tot[CBINS_MAX];
    const uint32_t* totals = totals_all + (size_t)blockIdx.x * nb;
    uint32_t* c = coarse_all + (size_t)blockIdx.x * coarse_stride;
    if (threadIdx.x < bins) {
        uint32_t sum = 0;
        for (uint32_t q = 0; q < (1u << fb); q++) sum += totals[(threadIdx.x << fb) + q];
        tot[threadIdx.x] = sum;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0, chunks = 0;
        for (uint32_t b = 0; b < bins; b++) {
            c[b] = run;
            c[(CBINS_MAX + 1) + b] = chunks;
            run += tot[b];
            chunks += (tot[b] + SUB - 1) / SUB;
        }
        c[bins] = run;
        c[(CBINS_MAX + 1) + bins] = chunks;
    }
}

// one LDS counting sort of up to SUB entries by a 6-bit (7-bit) key held in `key[]`, then the coalesced write-out:
// slot q of the sorted run goes to dst[gbase[key] + q - lstart[key]]
struct SortLds {
    uint32_t cnt[CBINS_MAX], lstart[CBINS_MAX + 1], gbase[CBINS_MAX];
    uint32_t sorted[SUB];
    uint8_t kid[SUB];
};

__device__ __forceinline__ void sort_scan(SortLds& S, uint32_t bins) {
    // exclusive scan of S.cnt over `bins` <= 256 keys by the first wave (four consecutive keys per lane);
    // S.lstart[CBINS_MAX] = the total
    if (threadIdx.x < 64) {
        const uint32_t k0 = threadIdx.x * 4;
        const uint32_t a0 = k0 < bins ? S.cnt[k0] : 0, a1 = k0 + 1 < bins ? S.cnt[k0 + 1] : 0;
        const uint32_t a2 = k0 + 2 < bins ? S.cnt[k0 + 2] : 0, a3 = k0 + 3 < bins ? S.cnt[k0 + 3] : 0;
        const uint32_t s = a0 + a1 + a2 + a3;
        uint32_t x = s;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t y = __shfl_up(x, off);
            if ((int)threadIdx.x >= off) x += y;
        }
        const uint32_t base = x - s;
        S.lstart[k0] = base;
        S.lstart[k0 + 1] = base + a0;
        S.lstart[k0 + 2] = base + a0 + a1;
        S.lstart[k0 + 3] = base + a0 + a1 + a2;
        if (threadIdx.x == 63) S.lstart[CBINS_MAX] = x;
    }
}

// level 1: a workgroup's FCHUNK scalars x nwin windows, SUB entry slots (SUB / FCHUNK windows) at a time: sorted in LDS by
// coarse bin and appended to the workgroup's range of every bin of `inter` (reserved by msm_recode_hist2_kernel)
__global__ __launch_bounds__(256) void msm_scatter1_kernel(const int16_t* __restrict__ digits_all, uint32_t n, uint32_t stride, uint32_t nwin,
                                                           uint32_t nb, uint32_t table_stride, const uint32_t* __restrict__ coarse_all,
                                                           uint32_t coarse_stride, uint32_t* __restrict__ inter_all, size_t inter_stride,
                                                           uint32_t fb) {
    __shared__ SortLds S;
    const uint32_t col = blockIdx.y;
    const int16_t* __restrict__ digits = digits_all + (size_t)col * nwin * stride;
    const uint32_t* __restrict__ chdr = coarse_all + (size_t)col * coarse_stride;
    const uint32_t* __restrict__ cbase = chdr + COARSE_WORDS + (size_t)blockIdx.x * CBINS_MAX;
    uint32_t* __restrict__ inter = inter_all + (size_t)col * inter_stride;
    const uint32_t bins = nb >> fb, fmask = (1u << fb) - 1;
    const uint32_t lo = blockIdx.x * FCHUNK, hi = min(n, lo + FCHUNK);
    if (threadIdx.x < bins) S.gbase[threadIdx.x] = chdr[threadIdx.x] + cbase[threadIdx.x];
    constexpr uint32_t WPS = SUB / FCHUNK;      // windows per sub-round
    constexpr uint32_t PER = SUB / 256;         // entry slots per lane and sub-round
    for (uint32_t w0 = 0; w0 < nwin; w0 += WPS) {
        if (threadIdx.x < CBINS_MAX) S.cnt[threadIdx.x] = 0;
        __syncthreads();
        uint32_t ent[PER], meta[PER];  // meta = key << 16 | rank, 0xffffffff = no entry
#pragma unroll
        for (uint32_t q = 0; q < PER; q++) {
            const uint32_t e = threadIdx.x + q * 256;  // slot: window w0 + e / FCHUNK, scalar lo + e % FCHUNK
            const uint32_t w = w0 + e / FCHUNK, i = lo + (e % FCHUNK);
            meta[q] = 0xffffffffu;
            if (w < nwin && i < hi) {
                const int32_t d = digits[(size_t)w * stride + i];
                if (d != 0) {
                    const uint32_t bkt = (uint32_t)(d < 0 ? -d : d) - 1;
                    const uint32_t key = bkt >> fb;
                    ent[q] = (w * table_stride + i) | ((bkt & fmask) << 24) | (d < 0 ? SIGN_BIT : 0);
                    meta[q] = (key << 16) | atomicAdd(&S.cnt[key], 1u);
                }
            }
        }
        __syncthreads();
        sort_scan(S, bins);
        __syncthreads();
#pragma unroll
        for (uint32_t q = 0; q < PER; q++)
            if (meta[q] != 0xffffffffu) {
                const uint32_t key = meta[q] >> 16, pos = S.lstart[key] + (meta[q] & 0xffffu);
                S.sorted[pos] = ent[q];
                S.kid[pos] = (uint8_t)key;
            }
        __syncthreads();
        const uint32_t total = S.lstart[CBINS_MAX];
        for (uint32_t q = threadIdx.x; q < total; q += 256) {
            const uint32_t key = S.kid[q];
            inter[S.gbase[key] + q - S.lstart[key]] = S.sorted[q];
        }
        __syncthreads();
        if (threadIdx.x < bins) S.gbase[threadIdx.x] += S.cnt[threadIdx.x];
    }
}

// level 2: one chunk (<= SUB entries) of one coarse bin into its 64 buckets; also the bucket padding (skip markers)
__global__ __launch_bounds__(256) void msm_scatter2_kernel(const uint32_t* __restrict__ inter_all, size_t inter_stride,
                                                           const uint32_t* __restrict__ coarse_all, uint32_t coarse_stride, uint32_t nb,
                                                           const uint32_t* __restrict__ totals_all,
                                                           const uint32_t* __restrict__ bucket_start_all, uint32_t* __restrict__ cursor_all,
                                                           uint32_t* __restrict__ entries_all, uint32_t fb, uint32_t pad,
                                                           size_t ent_stride, const uint8_t* __restrict__ delta_all) {
    // 13 / 14-bit plan: the bucket starts of all columns index one dense entry list (ent_stride = 0), ranges padded to
    // PAD entries (skip markers); wide path (delta_all != nullptr): column-local starts, one entry region per column, no
    // padding (pad = 1) — the first entry of every bucket carries WIDE_FLAG and the bucket's distance from the previous
    // non-empty one (msm_binscan_kernel) in its spare bits
    __shared__ SortLds S;
    __shared__ uint32_t s_bin, s_chunk;
    __shared__ uint32_t s_mark[CBINS_MAX];  // wide path: flag bits of a key's first entry when this chunk holds the bucket's first
    const uint32_t col = blockIdx.y;
    const uint32_t* __restrict__ inter = inter_all + (size_t)col * inter_stride;
    const uint32_t* __restrict__ chdr = coarse_all + (size_t)col * coarse_stride;
    const uint32_t* __restrict__ totals = totals_all + (size_t)col * nb;
    const uint32_t* __restrict__ bucket_start = bucket_start_all + (size_t)col * nb;
    uint32_t* __restrict__ cursor = cursor_all + (size_t)col * nb;
    uint32_t* __restrict__ entries = entries_all + (size_t)col * ent_stride;
    const uint32_t bins = nb >> fb, keys = 1u << fb;
    // this workgroup's share of the bucket padding (skip markers up to the next multiple of `pad`)
    for (uint32_t b = blockIdx.x * 256 + threadIdx.x; b < nb; b += gridDim.x * 256) {
        const uint32_t beg = bucket_start[b] + totals[b], end = bucket_start[b] + ((totals[b] + pad - 1) & ~(pad - 1));
        for (uint32_t q = beg; q < end; q++) entries[q] = SKIP_ENTRY;
    }
    const uint32_t* cpre = chdr + (CBINS_MAX + 1);
    if (blockIdx.x >= cpre[bins]) return;  // the grid is sized for the worst case
    if (threadIdx.x == 0) {
        uint32_t lo = 0, hi = bins;  // the bin whose chunk range holds blockIdx.x
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (cpre[mid] <= blockIdx.x) lo = mid;
            else hi = mid;
        }
        s_bin = lo;
        s_chunk = blockIdx.x - cpre[lo];
    }
    if (threadIdx.x < keys) S.cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t bin = s_bin;
    const uint32_t beg = chdr[bin] + s_chunk * SUB;
    const uint32_t end = min(chdr[bin + 1], beg + SUB);
    constexpr uint32_t PER = SUB / 256;
    uint32_t ent[PER], meta[PER];
#pragma unroll
    for (uint32_t q = 0; q < PER; q++) {
        const uint32_t p = beg + threadIdx.x + q * 256;
        meta[q] = 0xffffffffu;
        if (p < end) {
            const uint32_t e = inter[p];
            const uint32_t key = (e >> 24) & (keys - 1);
            ent[q] = e & ~((keys - 1) << 24);
            meta[q] = (key << 16) | atomicAdd(&S.cnt[key], 1u);
        }
    }
    __syncthreads();
    sort_scan(S, keys);
    if (threadIdx.x < keys) {
        const uint32_t cnt = S.cnt[threadIdx.x], b = bin * keys + threadIdx.x;
        const uint32_t before = cnt ? atomicAdd(&cursor[b], cnt) : 0;
        S.gbase[threadIdx.x] = bucket_start[b] + before;
        s_mark[threadIdx.x] = (delta_all && cnt && before == 0) ? (WIDE_FLAG | ((uint32_t)delta_all[(size_t)col * nb + b] << 24)) : 0;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t q = 0; q < PER; q++)
        if (meta[q] != 0xffffffffu) {
            const uint32_t key = meta[q] >> 16, pos = S.lstart[key] + (meta[q] & 0xffffu);
            S.sorted[pos] = ent[q];
            S.kid[pos] = (uint8_t)key;
        }
    __syncthreads();
    const uint32_t total = end - beg;
    for (uint32_t q = threadIdx.x; q < total; q += 256) {
        const uint32_t key = S.kid[q];
        entries[S.gbase[key] + q - S.lstart[key]] = S.sorted[q] | (q == S.lstart[key] ? s_mark[key] : 0u);
    }
}

#ifndef ZK_SCAT_T
#define ZK_SCAT_T 256
#endif
static constexpr uint32_t SCAT_T = ZK_SCAT_T;  // lanes of a scatter workgroup (FCHUNK / SCAT_T scalars per lane)
__global__ __launch_bounds__(SCAT_T) void msm_scatter_fixed_kernel(const int16_t* __restrict__ digits_all, uint32_t n, uint32_t stride,
                                                                uint32_t nwin, uint32_t nb, uint32_t table_stride,
                                                                const uint32_t* __restrict__ totals_all,
                                                                const uint32_t* __restrict__ bucket_start_all,
                                                                const uint32_t* __restrict__ blockbase_all,
                                                                uint32_t* __restrict__ entries) {
    extern __shared__ uint32_t lds[];  // nb cursors
    // column blockIdx.y of the launch: its digit planes, bucket totals / starts ([nb] is the next column's first
    // start, or the grand total) and reserved ranges
    const uint32_t col = blockIdx.y;
    const int16_t* __restrict__ digits = digits_all + (size_t)col * nwin * stride;
    const uint32_t* __restrict__ totals = totals_all + (size_t)col * nb;
    const uint32_t* __restrict__ bucket_start = bucket_start_all + (size_t)col * nb;
    const uint32_t* __restrict__ blockbase = blockbase_all + (size_t)col * gridDim.x * nb;
    for (uint32_t b = threadIdx.x; b < nb; b += SCAT_T) lds[b] = bucket_start[b] + blockbase[(size_t)blockIdx.x * nb + b];
    // this workgroup's share of the bucket padding (skip markers up to the next multiple of PAD)
    for (uint32_t b = blockIdx.x * SCAT_T + threadIdx.x; b < nb; b += gridDim.x * SCAT_T) {
        const uint32_t beg = bucket_start[b] + totals[b], end = bucket_start[b + 1];
        for (uint32_t q = beg; q < end; q++) entries[q] = SKIP_ENTRY;
    }
    __syncthreads();
    const uint32_t lo = blockIdx.x * FCHUNK, hi = min(n, lo + FCHUNK);
    constexpr uint32_t PER = FCHUNK / SCAT_T;  // scalars per thread
    // the digits of window w + 1 are loaded while those of window w are scattered (the loop is otherwise a
    // chain of load -> LDS atomic -> store latencies at two waves per SIMD)
    int32_t cur[PER], nxt[PER];
#pragma unroll
    for (uint32_t q = 0; q < PER; q++) {
        const uint32_t i = lo + threadIdx.x + q * SCAT_T;
        cur[q] = i < hi ? digits[i] : 0;
    }
    for (uint32_t w = 0; w < nwin; w++) {
        if (w + 1 < nwin) {
            const int16_t* dg = digits + (size_t)(w + 1) * stride;
#pragma unroll
            for (uint32_t q = 0; q < PER; q++) {
                const uint32_t i = lo + threadIdx.x + q * SCAT_T;
                nxt[q] = i < hi ? dg[i] : 0;
            }
        }
#pragma unroll
        for (uint32_t q = 0; q < PER; q++) {
            const int32_t d = cur[q];
            if (d == 0) continue;
            const uint32_t i = lo + threadIdx.x + q * SCAT_T;
            const uint32_t pos = atomicAdd(&lds[(d < 0 ? -d : d) - 1], 1u);
            entries[pos] = (w * table_stride + i) | (d < 0 ? SIGN_BIT : 0);
        }
#pragma unroll
        for (uint32_t q = 0; q < PER; q++) cur[q] = nxt[q];
    }
}

// exclusive scan of the bucket sizes, each rounded up to a multiple of PAD (so that neither an
// accumulate lane nor a first-level gather lane straddles two buckets): out[0..m], out[m] = padded total = counts[0]
__global__ __launch_bounds__(1024) void msm_scan_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                        uint32_t m, uint32_t* __restrict__ counts) {
    __shared__ uint32_t part[1024];
    const uint32_t chunk = (m + 1023) / 1024;
    const uint32_t lo = min(m, threadIdx.x * chunk);
    const uint32_t hi = min(m, lo + chunk);
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; i++) sum += (in[i] + PAD - 1) & ~(PAD - 1);
    part[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        uint32_t v = (threadIdx.x >= d) ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;
    for (uint32_t i = lo; i < hi; i++) {
        const uint32_t h = (in[i] + PAD - 1) & ~(PAD - 1);
        out[i] = run;
        run += h;
    }
    if (threadIdx.x == 1023) {
        out[m] = part[1023];
        counts[0] = part[1023];
    }
}

// fill the padding at the end of every bucket with skip markers
__global__ void msm_pad_kernel(const uint32_t* __restrict__ totals, const uint32_t* __restrict__ bucket_start, uint32_t m,
                               uint32_t* __restrict__ entries) {
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= m) return;
    const uint32_t beg = bucket_start[b] + totals[b], end = bucket_start[b + 1];
    for (uint32_t p = beg; p < end; p++) entries[p] = SKIP_ENTRY;
}

// ------------------------------------------------------------ accumulate ---

// Every lane sums one aligned segment of SEG0 entries; bucket ranges are padded to multiples
// of SEG0, so a segment lies inside ONE bucket and yields one partial sum ("slot").
// four waves per SIMD (128 registers; the few values that do not fit live in scratch words of the rare paths): needed by the
// serial multiply-add columns of the addition (field29.hip.h mul29s), neutral otherwise
#ifndef ZK_ACC_WAVES
#define ZK_ACC_WAVES 4
#endif
// One segment.  SAFE: every addition tests for the identity as an operand and for the exceptional cases (same x as the running
// sum: a doubling or a cancellation), which are redone on the general formulas — exact for any input.  Otherwise no test at
// all (4 % faster: it is the branches around the fallback more than the instructions): the caller vouches that no base is the
// identity and checks the segment's ZZ afterwards.
template <bool SAFE>
__device__ __forceinline__ G1X29 accumulate_segment(const uint32_t* __restrict__ e, const G1Affine* __restrict__ bases) {
    G1X29 acc;
    acc.inf = true;
    for (uint32_t k = 0; k < SEG0; k++) {
        const uint32_t y = e[k];
        if (y == SKIP_ENTRY) continue;  // padding at the end of a bucket
        G1Affine p = affine_load(bases + (y & ~SIGN_BIT));
        if (SAFE && affine_is_identity(p)) continue;
        if (y & SIGN_BIT) p.y = fe_neg(p.y);
        if (!g1x29_add_affine<SAFE>(acc, p.x, p.y)) {
            // same x as the running sum (doubling or cancellation): the general formulas, rarely
            G1X s = g1x29_to_std(acc);
            g1x_add_affine(s, p.x, p.y);
            acc = g1x29_from_std(s);
        }
    }
    return acc;
}

// Checked kernel: exact for any bases (arbitrary bases of the fine-grained seam; an SRS that holds the identity).
#if ZK_ACC_WAVES
__attribute__((amdgpu_waves_per_eu(ZK_ACC_WAVES, ZK_ACC_WAVES)))
#endif
__global__ __launch_bounds__(64) void msm_accumulate_kernel(const uint32_t* __restrict__ entries,
                                                            const G1Affine* __restrict__ bases,
                                                            const uint32_t* __restrict__ counts,
                                                            G1X29S* __restrict__ slot_pt) {
    const uint32_t total = counts[0];  // multiple of SEG0
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t * SEG0 >= total) return;
    // the running sum lives on the carry-free 29-bit-limb field (ec29.hip.h); bases are read in
    // their standard memory form, the slot is written in the internal one (the reduction tails stay on that field)
    g1x29_store(slot_pt + t, accumulate_segment<true>(entries + (size_t)t * SEG0, bases));
}

// Unchecked kernel for a basis without the identity (the resident SRS): no test at all in the loop, and none of the
// fallback code in the kernel.  An exceptional step leaves ZZ = 0 (ZZ is the product of the squared x-differences and p is
// prime): such segments are listed (counts[1], redo[]) and msm_accumulate_redo_kernel, which always follows, redoes them with
// the checked loop — with distinct bases a handful of segments per MSM, if any.
#if ZK_ACC_WAVES
__attribute__((amdgpu_waves_per_eu(ZK_ACC_WAVES, ZK_ACC_WAVES)))
#endif
__global__ __launch_bounds__(64) void msm_accumulate_fast_kernel(const uint32_t* __restrict__ entries,
                                                                 const G1Affine* __restrict__ bases,
                                                                 uint32_t* __restrict__ counts, uint32_t* __restrict__ redo,
                                                                 G1X29S* __restrict__ slot_pt) {
    const uint32_t total = counts[0];  // multiple of SEG0
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t * SEG0 >= total) return;
    const G1X29 acc = accumulate_segment<false>(entries + (size_t)t * SEG0, bases);
    if (!acc.inf && is_zero29(acc.zz)) redo[atomicAdd(&counts[1], 1u)] = t;  // at most one entry per segment: redo[] has one word each
    g1x29_store(slot_pt + t, acc);
}
__global__ __launch_bounds__(64) void msm_accumulate_redo_kernel(const uint32_t* __restrict__ entries,
                                                                 const G1Affine* __restrict__ bases,
                                                                 const uint32_t* __restrict__ counts, const uint32_t* __restrict__ redo,
                                                                 G1X29S* __restrict__ slot_pt) {
    const uint32_t m = counts[1];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
        const uint32_t t = redo[i];
        g1x29_store(slot_pt + t, accumulate_segment<true>(entries + (size_t)t * SEG0, bases));
    }
}

// start-of-MSM reset in one launch: bucket parts = identity, bucket totals = 0, counts = 0
__global__ void msm_clear_kernel(G1X29S* __restrict__ p, uint32_t m, uint32_t* __restrict__ totals, uint32_t nt,
                                 uint32_t* __restrict__ counts, uint32_t* __restrict__ cursor, uint32_t ncur,
                                 uint32_t* __restrict__ coarse, uint32_t coarse_stride, uint32_t ncols) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) g1x29_store(p + i, g1x29_identity());
    if (i < nt) totals[i] = 0;
    if (i < 4 * (ncols + 1)) counts[i] = 0;
    if (i < ncur) cursor[i] = 0;  // second-level write cursors (two-level sort)
    if (i < ncols * CBINS_MAX) coarse[(size_t)(i / CBINS_MAX) * coarse_stride + 2 * (CBINS_MAX + 1) + (i % CBINS_MAX)] = 0;  // first-level append cursors
}

#ifdef ZK_TAIL_TRACE  // tools/ubench_tail.hip: where a bit-sum workgroup spends its time (100 MHz wall clock stamps)
__device__ unsigned long long zk_tail_trace[16];
__device__ unsigned long long zk_wg_trace[3][8192];  // per-workgroup begin / middle / end of the last traced kernel
#define ZK_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) zk_tail_trace[i] = wall_clock64(); } while (0)
#define ZK_WG_STAMP(i) do { if (blockIdx.x < 8192 && threadIdx.x == 0) zk_wg_trace[i][blockIdx.x] = wall_clock64(); } while (0)
#else
#define ZK_STAMP(i) do { } while (0)
#define ZK_WG_STAMP(i) do { } while (0)
#endif

// ---- reduction tails.  The partial sums stay on the carry-free 29-bit-limb field (ec29.hip.h: 3 300 instructions per
// general XYZZ addition with the products inlined, against 4 600 through out-of-line 8 x 32-bit products, and no dependent
// carry chains — these kernels run at one or two waves per SIMD, where a chained product is latency-bound).  Every kernel
// is shaped so that it has ONE inlined addition (a loop that fetches its operand from memory, from a shuffle or from LDS
// and then adds): three copies of the addition would not fit the instruction cache.

// First-level gather: every lane sums GA consecutive slots serially (same bucket by alignment):
// dense lanes, no idle tree steps — this is where most of the slot additions happen.
__global__ __launch_bounds__(64) void msm_gather1_kernel(const G1X29S* __restrict__ slot_pt, const uint32_t* __restrict__ counts,
                                                         G1X29S* __restrict__ partial) {
    const uint32_t t = blockIdx.x * 64 + threadIdx.x;
    if ((size_t)t * PAD >= counts[0]) return;
    const G1X29S* src = slot_pt + (size_t)t * GA;
    G1X29 acc = g1x29_load(src);
#pragma unroll 1
    for (uint32_t k = 1; k < GA; k++) {
        const G1X29 v = g1x29_load(src + k);
        g1x29_add<ZK_TAIL_SER>(acc, v);
    }
    g1x29_store(partial + t, acc);
}

// Parts of bucket b that the second-level gather writes and the bit sums read — ONE definition for both kernels: the
// parts are not reset between MSMs (msm_run), so a disagreement would make the bit sums read a previous MSM's sums.
// -DZK_MSM_POISON fills the parts with a non-point before every MSM to catch exactly that.
__device__ __forceinline__ uint32_t msm_used_parts(const uint32_t* __restrict__ bucket_start, uint32_t b, uint32_t parts) {
    const uint32_t len = bucket_start[b + 1] / PAD - bucket_start[b] / PAD;
    return min(parts, (len + GSHARE - 1) / GSHARE);
}

// Second level: one LANES-lane group per (bucket b, part p) sums the p-th share of the bucket's
// first-level partials ([start_b / PAD, start_{b+1} / PAD) — contiguous, all of bucket b): lanes
// stride over the share, then a shuffle tree.  A bucket uses ceil(partials / (4 * LANES)) parts
// (at most `parts`); the others stay identity.
template <uint32_t LANES>
__global__ __launch_bounds__(256) void msm_gather_kernel(const uint32_t* __restrict__ bucket_start,
                                                         const G1X29S* __restrict__ partial, uint32_t parts, uint32_t ngroups,
                                                         G1X29S* __restrict__ part) {
    const uint32_t gid = (blockIdx.x * 256 + threadIdx.x) / LANES;
    const uint32_t lane = threadIdx.x & (LANES - 1);
    G1X29 acc = g1x29_identity();
    bool active = false;
    uint32_t b = 0, p = 0, s = 0, a1 = 0;
    if (gid < ngroups) {
        // part-major: the groups of part 0 (the only one most buckets use) are adjacent, so their waves are
        // full and the waves of the unused parts exit at once
        const uint32_t nbk = ngroups / parts;
        p = gid / nbk;
        b = gid - p * nbk;
        const uint32_t s0 = bucket_start[b] / PAD, s1 = bucket_start[b + 1] / PAD;
        const uint32_t len = s1 - s0;
        const uint32_t used = msm_used_parts(bucket_start, b, parts);
        if (p < used) {
            active = true;
            const uint32_t share = (len + used - 1) / used;
            const uint32_t a0 = s0 + p * share;
            a1 = min(s1, a0 + share);
            s = a0 + lane;
        }
    }
    if (!__any(active)) return;  // wave-uniform: no group of this wave has work
    // the serial part (lanes stride over the share) and the shuffle tree feed the same addition
    int off = LANES >> 1;
    ZK_STAMP(8);
    ZK_WG_STAMP(0);
#pragma unroll 1
    for (;;) {
        G1X29 v;
        bool have;
        if (__any(active && s < a1)) {  // wave-uniform
            have = active && s < a1;
            if (have) v = g1x29_load(partial + s);
            s += LANES;
        } else {
            if (off == 0) break;
            if (off == (int)(LANES >> 1)) {
                ZK_STAMP(9);
                ZK_WG_STAMP(1);
            }
            v = g1x29_shfl_down(acc, off);  // every lane of the wave takes part in the shuffles
            have = (int)lane < off;
            off >>= 1;
        }
        if (have) g1x29_add<ZK_TAIL_SER>(acc, v);
    }
    ZK_STAMP(10);
    ZK_WG_STAMP(2);
    if (active && lane == 0) g1x29_store(part + (size_t)b * parts + p, acc);
}

// ---------------------------------------------------------------- reduce ---
// sum_j j * B_j = sum_t 2^t * G_t with G_t = sum of the buckets whose multiplier j has
// bit t set: c tree reductions per bucket set instead of 2^(c-1) scalar multiplications.
// grid = slices * c * split workgroups of THREADS lanes, one multiplier per lane: the shape follows the
// bucket count (2^(c-1) / 2 multipliers per bit: 4 x 512 lanes at c = 13, 1 x 128 at c = 9), so that the
// tree is no deeper than the data and small bucket sets do not launch idle waves.  The host adds the
// `split` partials of a bit and runs the c-term Horner (on the standard form: the one lane that writes a
// bit sum converts it).
static constexpr uint32_t BITSUM_MAX_SPLIT = 4;
// One wave per workgroup.  Measured on the 4 x 2048 multipliers of a 13-bit window (tools/ubench_bitsum.hip): workgroups of
// 512 lanes (one multiplier per lane, a cross-wave step through LDS) 205 us, 256 lanes 140, 128 lanes 113, 64 lanes 107 —
// the hardware packs the waves of a workgroup two or three to a SIMD even on an idle chip, each tree step then costs two or
// three additions, and the others wait at the barrier; a lone wave per workgroup gets a SIMD to itself.
static uint32_t bitsum_threads(uint32_t nb) {
    (void)nb;
    return 64;
}
static uint32_t bitsum_split(uint32_t nb) {
    uint32_t s = (nb >> 1) / 512;
    if (s < 1) s = 1;
    if (s > BITSUM_MAX_SPLIT) s = BITSUM_MAX_SPLIT;
    return s;
}
template <uint32_t THREADS>
__global__ __launch_bounds__(THREADS) void msm_bitsum_kernel(const G1X29S* __restrict__ part, uint32_t parts, uint32_t nb,
                                                             uint32_t c, uint32_t split, const uint32_t* __restrict__ bucket_start,
                                                             G1X* __restrict__ out) {
    __shared__ G1X29S sh[THREADS / 64];
    const uint32_t q = blockIdx.x % split;
    const uint32_t st = blockIdx.x / split;
    const uint32_t slice = st / c, t = st - slice * c;
    const uint32_t wave = threadIdx.x >> 6;
    G1X29 acc = g1x29_identity();
    // the multipliers j in [1, nb] with bit t set, enumerated densely (no lane idles on a clear bit):
    // t < c-1: j = i with a 1 inserted at bit t, i < nb/2;  t = c-1: j = nb only
    const uint32_t items = t + 1 < c ? nb >> 1 : 1;
    uint32_t i = q * THREADS + threadIdx.x, k = 0;
    // parts of a bucket the gather kernel wrote (the others are identity and not worth a round trip to memory)
    const auto used_parts = [&](uint32_t b) { return msm_used_parts(bucket_start, b, parts); };
    const auto multiplier = [&](uint32_t ii) { return t + 1 < c ? (((ii >> t) << (t + 1)) | (1u << t) | (ii & ((1u << t) - 1))) : nb; };
    uint32_t used = 0;
    ZK_STAMP(0);
    while (i < items && (used = used_parts(slice * nb + multiplier(i) - 1)) == 0) i += THREADS * split;
    ZK_STAMP(1);
    // stage 0: the lane's multipliers (serial), then a shuffle tree over the wave; stage 1 (wave 0 only): the
    // per-wave sums from LDS and a shuffle tree over them.  One loop, one addition.
    int off = 32;
    uint32_t lanes = 64;
#pragma unroll 1
    for (uint32_t stage = 0;; stage++) {
#pragma unroll 1
        for (;;) {
            G1X29 v;
            bool have;
            if (stage == 0 && __any(i < items)) {  // wave-uniform
                have = i < items;
                if (have) {
                    v = g1x29_load(part + ((size_t)slice * nb + (multiplier(i) - 1)) * parts + k);
                    if (++k == used) {
                        k = 0;
                        i += THREADS * split;
                        while (i < items && (used = used_parts(slice * nb + multiplier(i) - 1)) == 0) i += THREADS * split;
                    }
                }
            } else {
                if (off == 0) break;
                if (stage == 0 && off == 32) ZK_STAMP(2);
                v = g1x29_shfl_down(acc, off);
                have = (threadIdx.x & (lanes - 1)) < (uint32_t)off;
                off >>= 1;
            }
            if (have) g1x29_add<ZK_TAIL_SER>(acc, v);
        }
        ZK_STAMP(3 + 2 * stage);
        if (THREADS == 64 || stage == 1) break;
        if ((threadIdx.x & 63) == 0) g1x29_store(sh + wave, acc);
        __syncthreads();
        ZK_STAMP(4);
        if (wave != 0) return;
        acc = (threadIdx.x < THREADS / 64) ? g1x29_load(sh + threadIdx.x) : g1x29_identity();
        lanes = THREADS / 64;
        off = (int)(THREADS / 128);
    }
    if (threadIdx.x == 0) {
        G1X r = G1X::identity();
        if (!acc.inf) {
            r.x = internal_to_std_call(acc.x);
            r.y = internal_to_std_call(acc.y);
            r.zz = internal_to_std_call(acc.zz);
            r.zzz = internal_to_std_call(acc.zzz);
        }
        g1x_store(out + blockIdx.x, r);
        ZK_STAMP(6);
    }
}

// ================================================================== wide path ==
// Windows of 15 / 16 bits (fixed-base mode): 17 / 16 bucket additions per scalar instead of 20 at 13 bits, paid for with
// 16384 / 32768 buckets per column.  What changes against the 13-bit plan above:
//   * the two-level sort splits a bucket index into an 8-bit coarse bin and a 7-bit fine key (128 buckets per bin); the
//     digits kernel counts coarse bins only, the per-bucket totals are counted from the binned intermediate list
//     (32768 LDS counters per workgroup would cost 6.7 M global atomics per 2^19 column);
//   * every column owns a region of the entry / slot / part lists, and all scans are local to a coarse bin;
//   * NO PADDING: the entry list is dense.  A lane of the accumulation sums WL consecutive entries whatever buckets they
//     fall in: the first entry of every bucket carries a flag, at which the lane stores its running sum (one "slot" per
//     (lane, bucket) pair: slot index = lane + bucket, unique and ordered along the staircase of the pairs) and restarts.
//     A restart costs a few moves because the window tables of this path hold the points in the accumulation's internal
//     form (x * 2^261: g1x29_add_affine<.., INTERNAL>).  Against 16-entry segments padded per bucket: no padding lanes
//     (3 %), no skip markers to write, and the run length WL is free to choose (measured below);
//   * the reduction tail is shaped for many small buckets: (T1) one lane per "part" of at most WCAP slots sums it
//     serially, lanes of the same bucket inside a wave are joined by a segmented shuffle tree; (T2) the bucket matrix
//     [rows = nb / 256][256] is summed along its rows and along its columns, one wave each —
//     sum_b (b + 1) B_b = 256 sum_h h R_h + sum_l (l + 1) C_l — and (T3) the short weighted sums over h and l + 1 are
//     taken bit by bit (log2(rows) + 9 tree reductions per column); the host runs the 16-step Horner.
// slots per part (T1's serial run), a per-pass parameter.  8 everywhere: lane-serial additions are the cheap ones (every lane
// busy); shorter runs for a lone column — whose tail is exposed latency — were measured (tools/single_ab.py, k = 19 single
// proof): 8: 12.37-12.41 ms, 4: 12.41-12.49, 2: 12.55-12.60 (more waves and more tree levels cost what the shorter chain saves)
#ifndef ZK_WCAP_BATCH
#define ZK_WCAP_BATCH 8
#endif
static constexpr uint32_t WCAP_MIN = 2, WCAP_BATCH = ZK_WCAP_BATCH;  // WCAP_MIN sizes the part lists
#ifndef ZK_WCAP_ONE
#define ZK_WCAP_ONE 8
#endif
static inline uint32_t wcap_for(uint32_t batch) { return batch == 1 ? (uint32_t)ZK_WCAP_ONE : WCAP_BATCH; }
static constexpr uint32_t WIDE_SUMS = 20; // bit sums per column handed to the host: 9 column bits, then up to 8 row bits (65 536 buckets)
#ifndef ZK_WL
#define ZK_WL 16
#endif
// entries per accumulation lane.  Measured with whole proofs, two pipelines in flight (tools/bench_ab.sh, proofs/s | single
// proof ms): 8: 90.0 | 12.9, 11: 91.4 | 12.9, 12: 92.8 | 12.5, 16: 95.0 | 12.2, 32: 91.4 | 12.8, 48: 85.8 | 13.7, 64: 84.4 | 13.9
// — short runs mean more slots for the reduction tail to add up, long runs fewer, longer waves that share the chip badly
// with the other kernels in flight (a 2^19 column is 8192 waves of 16 additions)
static constexpr uint32_t WL = ZK_WL;
// slots of a bucket whose cnt > 0 entries start at position s of the column's entry list: one per lane that holds some of them
__device__ __forceinline__ uint32_t wide_slot_count(uint32_t s, uint32_t cnt) { return cnt ? (s + cnt - 1) / WL - s / WL + 1 : 0; }

// ---- the wide path's head (round 4): digits in registers, windows of 15 / 16 / 17 bits, table indexes of 24 .. 26 bits ----
// What changed against the round-3 head (digit planes in int16, a kernel per scan):
//   * signed digits without a carry chain: with K = sum_w 2^(c - 1 + w c) the unsigned windows u_w of s + K give
//     d_w = u_w - 2^(c-1) in [-2^(c-1), 2^(c-1)) with sum_w d_w 2^(w c) = s — the same digits as the carry recoding (the
//     expansion with all digits in that range is unique), but window w needs nothing from window w - 1.  So the digits are
//     recomputed from the scalar wherever they are needed and the int16 digit planes (and their 16-bit limit) are gone;
//   * 17-bit windows: 15 bucket additions per scalar instead of 16 (65 536 buckets per column, 512 coarse bins);
//   * an entry's table index takes ib = 24 .. 26 bits (16 windows x 2^21 points need 25), the fine key / distance fields
//     the rest: fb = 31 - ib fine bits (128 / 64 / 32 buckets per coarse bin), 30 - ib distance bits;
//   * the coarse scan is the prologue of the first scatter (every workgroup scans the <= 512 bin totals itself, workgroup
//     0 publishes the header), the reset of the counters is the epilogue of the last tail kernel.
static constexpr uint32_t WCB = 512;                  // coarse bins at most (65 536 buckets / 128, or 32 768 / 64)
// per column: bin starts, chunk prefix, (unused), (unused), part regions — WCB + 1 words each — then the append cursors, ONE
// PER 128-BYTE LINE: every workgroup of H1 ends with a returning atomic per bin on them (131 K atomics on 256 words per 2^19
// column); side by side they would all land on 8 cache lines
#ifndef ZK_WCUR_STRIDE
#define ZK_WCUR_STRIDE 32
#endif
static constexpr uint32_t WCUR = ZK_WCUR_STRIDE;      // words between two append cursors
static constexpr uint32_t WCUR0 = 5 * (WCB + 1);      // first cursor
static constexpr uint32_t WHDR = WCUR0 + WCB * WCUR;  // words of the header; the workgroups' reserved bases [blocks][WCB] follow
static constexpr uint32_t WSUB = 256 * 17;            // entries of a scatter sub-round: 256 scalars x all their windows (<= 17)

struct WideGeo {
    uint32_t c, nwin, nb;   // window bits, windows, buckets per column
    uint32_t ib, fb, bins;  // table index bits, fine key bits, coarse bins = nb >> fb
    uint32_t K[8];          // the recoding bias sum_w 2^(c - 1 + w c)
};

// u = (s + K) as 9 words (canonical s: the Montgomery image is taken off here)
__device__ __forceinline__ void wide_biased(const Fr& mont, const WideGeo& g, uint32_t (&u)[9]) {
    const Fr s = fe_from_mont(mont);
    uint64_t cy = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        cy += (uint64_t)s.v[k] + g.K[k];
        u[k] = (uint32_t)cy;
        cy >>= 32;
    }
    u[8] = (uint32_t)cy;  // 0: s + K < 2^256 (see msm_wide_geo)
}
// digit w of the biased scalar: C, w compile-time -> static register indexing
template <uint32_t C>
__device__ __forceinline__ int32_t wide_digit(const uint32_t (&u)[9], uint32_t w) {
    const uint32_t bit = w * C, word = bit >> 5, off = bit & 31;
    const uint64_t two = (uint64_t)u[word] | ((uint64_t)u[word + 1] << 32);
    return (int32_t)((uint32_t)(two >> off) & ((1u << C) - 1)) - (int32_t)(1u << (C - 1));
}

// exclusive scan of f(in[k]), k < bins <= 512, by ONE wave (eight consecutive bins per lane): out[k], out[bins] = total
template <class F>
__device__ __forceinline__ void wide_wave_scan(const uint32_t* in, uint32_t bins, uint32_t* out, F f) {
    const uint32_t lane = threadIdx.x & 63, k0 = lane * 8;
    uint32_t v[8], s = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        v[j] = k0 + j < bins ? f(in[k0 + j]) : 0;
        s += v[j];
    }
    uint32_t x = s;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t y = __shfl_up(x, off);
        if ((int)lane >= off) x += y;
    }
    uint32_t run = x - s;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        if (k0 + j < bins) out[k0 + j] = run;
        run += v[j];
    }
    if (lane == 63) out[bins] = x;
}

// H1: coarse histogram of a workgroup's FCHUNK scalars (all windows) and the reservation of its range inside every coarse
// bin of `inter`: one returning atomic per non-empty bin on the append cursors, which end up holding the bins' totals
template <uint32_t C>
__global__ __launch_bounds__(256) void msm_whist_kernel(MsmBatch batch, uint32_t n, WideGeo g, uint32_t* __restrict__ coarse_all,
                                                        uint32_t coarse_stride, uint32_t* __restrict__ counts) {
    __shared__ uint32_t hist[WCB];
    constexpr uint32_t NWIN = 254 / C + 1;
    const uint32_t col = blockIdx.y;
    const Fr* __restrict__ scalars = batch.s[col];
    uint32_t* __restrict__ chdr = coarse_all + (size_t)col * coarse_stride;
    if (col == 0 && blockIdx.x == 0 && threadIdx.x == 0) counts[1] = 0;  // this pass's redo count
    for (uint32_t b = threadIdx.x; b < g.bins; b += 256) hist[b] = 0;
    __syncthreads();
    const uint32_t lo = blockIdx.x * FCHUNK, hi = min(n, lo + FCHUNK);
    constexpr uint32_t PERL = FCHUNK / 256;
    Fr sv[PERL];  // the lane's scalars, all loads in flight together
#pragma unroll
    for (uint32_t q = 0; q < PERL; q++) {
        const uint32_t i = lo + threadIdx.x + q * 256;
        sv[q] = i < hi ? fe_load(scalars + i) : Fr::zero();
    }
#pragma unroll
    for (uint32_t q = 0; q < PERL; q++) {
        if (lo + threadIdx.x + q * 256 >= hi) break;
        uint32_t u[9];
        wide_biased(sv[q], g, u);
#pragma unroll
        for (uint32_t w = 0; w < NWIN; w++) {
            const int32_t d = wide_digit<C>(u, w);
            if (d != 0) atomicAdd(&hist[((uint32_t)(d < 0 ? -d : d) - 1) >> g.fb], 1u);
        }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < g.bins; b += 256) {
        const uint32_t sum = hist[b];
        chdr[WHDR + (size_t)blockIdx.x * WCB + b] = sum ? atomicAdd(&chdr[WCUR0 + b * WCUR], sum) : 0;
    }
}

struct WSortLds {
    uint32_t cnt[WCB], lstart[WCB + 1], gbase[WCB];
    uint32_t sorted[WSUB];
    uint16_t kid[WSUB];
};

// H2: level 1 of the sort.  Prologue: the bins' places in `inter` from the totals H1 left in the append cursors (every
// workgroup scans them itself; workgroup 0 of a column also publishes the header the later kernels read: bin starts, the
// chunk prefix of level 2, the bins' part regions — sized for the worst case, every bucket of a bin one slot and one part
// more than its share — and counts[4 col] = entries, counts[4 col + 2] = the end of the last part region).  Then, 256
// scalars at a time: digits in registers, entries sorted in LDS by coarse bin, runs appended to the workgroup's ranges.
template <uint32_t C>
__global__ __launch_bounds__(256) void msm_wscatter1_kernel(MsmBatch batch, uint32_t n, WideGeo g, uint32_t table_stride,
                                                            uint32_t* __restrict__ coarse_all, uint32_t coarse_stride,
                                                            uint32_t* __restrict__ inter_all, size_t inter_stride,
                                                            uint32_t* __restrict__ counts, uint32_t WCAP) {
    __shared__ WSortLds S;
    constexpr uint32_t NWIN = 254 / C + 1;
    constexpr uint32_t CB = WCB + 1;
    const uint32_t col = blockIdx.y;
    const Fr* __restrict__ scalars = batch.s[col];
    uint32_t* __restrict__ chdr = coarse_all + (size_t)col * coarse_stride;
    const uint32_t* __restrict__ cbase = chdr + WHDR + (size_t)blockIdx.x * WCB;
    uint32_t* __restrict__ inter = inter_all + (size_t)col * inter_stride;
    const uint32_t bins = g.bins, keys = 1u << g.fb;
    for (uint32_t b = threadIdx.x; b < bins; b += 256) S.cnt[b] = chdr[WCUR0 + b * WCUR];  // the bins' totals (H1's cursors)
    __syncthreads();
    const uint32_t* tot = S.cnt;
    if (threadIdx.x < 64) {
        wide_wave_scan(tot, bins, S.lstart, [](uint32_t t) { return t; });
        if (blockIdx.x == 0) {
            wide_wave_scan(tot, bins, chdr, [](uint32_t t) { return t; });
            wide_wave_scan(tot, bins, chdr + CB, [](uint32_t t) { return (t + SUB - 1) / SUB; });
            wide_wave_scan(tot, bins, chdr + 4 * CB, [=](uint32_t t) { return t ? (t / WL + 2 * keys + WCAP - 1) / WCAP + keys : 0u; });
        }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < bins; b += 256) S.gbase[b] = S.lstart[b] + cbase[b];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        counts[4 * col] = S.lstart[bins];
        counts[4 * col + 2] = chdr[4 * CB + bins];  // written by this wave above (same lane 63 -> memory; read back after the barrier)
    }
    const uint32_t lo = blockIdx.x * FCHUNK, hi = min(n, lo + FCHUNK);
    Fr next = lo + threadIdx.x < hi ? fe_load(scalars + lo + threadIdx.x) : Fr::zero();
    for (uint32_t i0 = lo; i0 < hi; i0 += 256) {
        __syncthreads();  // gbase / the previous round's lstart, kid, sorted are free
        for (uint32_t b = threadIdx.x; b < bins; b += 256) S.cnt[b] = 0;
        __syncthreads();
        const uint32_t i = i0 + threadIdx.x;
        const Fr cur = next;  // the next round's scalar is loaded under this round's sort
        if (i + 256 < hi) next = fe_load(scalars + i + 256);
        uint32_t ent[NWIN], meta[NWIN];  // meta = key << 16 | rank, 0xffffffff = no entry
#pragma unroll
        for (uint32_t w = 0; w < NWIN; w++) meta[w] = 0xffffffffu;
        if (i < hi) {
            uint32_t u[9];
            wide_biased(cur, g, u);
#pragma unroll
            for (uint32_t w = 0; w < NWIN; w++) {
                const int32_t d = wide_digit<C>(u, w);
                if (d != 0) {
                    const uint32_t bkt = (uint32_t)(d < 0 ? -d : d) - 1;
                    const uint32_t key = bkt >> g.fb;
                    ent[w] = (w * table_stride + i) | ((bkt & (keys - 1)) << g.ib) | (d < 0 ? SIGN_BIT : 0);
                    meta[w] = (key << 16) | atomicAdd(&S.cnt[key], 1u);
                }
            }
        }
        __syncthreads();
        if (threadIdx.x < 64) wide_wave_scan(S.cnt, bins, S.lstart, [](uint32_t t) { return t; });
        __syncthreads();
#pragma unroll
        for (uint32_t w = 0; w < NWIN; w++)
            if (meta[w] != 0xffffffffu) {
                const uint32_t key = meta[w] >> 16, pos = S.lstart[key] + (meta[w] & 0xffffu);
                S.sorted[pos] = ent[w];
                S.kid[pos] = (uint16_t)key;
            }
        __syncthreads();
        const uint32_t total = S.lstart[bins];
        for (uint32_t q = threadIdx.x; q < total; q += 256) {
            const uint32_t key = S.kid[q];
            inter[S.gbase[key] + q - S.lstart[key]] = S.sorted[q];
        }
        __syncthreads();
        for (uint32_t b = threadIdx.x; b < bins; b += 256) S.gbase[b] += S.cnt[b];
    }
}

// per-bucket totals from the sorted intermediate list: one workgroup per chunk (<= SUB entries) of a coarse bin — the
// decomposition of the second sort level — counts its entries per fine key in LDS and adds the counts to totals[]
// (zero at the start of a pass).  Chunks, not whole bins: witness-like columns put most of their entries into a few bins
// (a permuted lookup column at k = 19: 400 K entries in bin 0 — one workgroup needed 200 us for them).
__global__ __launch_bounds__(256) void msm_wfinehist_kernel(const uint32_t* __restrict__ inter_all, size_t inter_stride,
                                                            const uint32_t* __restrict__ coarse_all, uint32_t coarse_stride, WideGeo g,
                                                            uint32_t* __restrict__ totals_all, uint32_t* __restrict__ next_totals,
                                                            uint32_t* __restrict__ next_cursor, uint32_t next_cols) {
    __shared__ uint32_t hist[128];
    __shared__ uint32_t s_bin, s_chunk;
    const uint32_t col = blockIdx.y, keys = 1u << g.fb;
    const uint32_t* __restrict__ inter = inter_all + (size_t)col * inter_stride;
    const uint32_t* __restrict__ chdr = coarse_all + (size_t)col * coarse_stride;
    const uint32_t* cpre = chdr + (WCB + 1);
    {
        // the NEXT pass's per-bucket counters — the other set: nothing of this pass touches it — shared among this kernel's
        // (worst-case many) workgroups (next_cols: the columns the last pass on that set used: it may have been a wider batch than
        // this one).  In H1, whose lanes all have a scalar to wait for, the same stores cost 13 us
        const uint32_t step = gridDim.x * 256;
        for (uint32_t cc = col; cc < next_cols; cc += gridDim.y)
            for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < g.nb; i += step) {
                next_totals[(size_t)cc * g.nb + i] = 0;
                next_cursor[(size_t)cc * g.nb + i] = 0;
            }
    }
    if (blockIdx.x == gridDim.x - 1) {
        // (one block more than the worst case of chunks is launched for this)  The append cursors of the coarse bins have been
        // read by the first scatter: back to zero for the next pass's H1
        uint32_t* cur = const_cast<uint32_t*>(chdr) + WCUR0;
        for (uint32_t b = threadIdx.x; b < WCB; b += 256) cur[b * WCUR] = 0;
    }
    if (blockIdx.x >= cpre[g.bins]) return;  // the grid is sized for the worst case
    if (threadIdx.x == 0) {
        uint32_t lo = 0, hi = g.bins;  // the bin whose chunk range holds blockIdx.x
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (cpre[mid] <= blockIdx.x) lo = mid;
            else hi = mid;
        }
        s_bin = lo;
        s_chunk = blockIdx.x - cpre[lo];
    }
    if (threadIdx.x < keys) hist[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t bin = s_bin;
    const uint32_t beg = chdr[bin] + s_chunk * SUB;
    const uint32_t end = min(chdr[bin + 1], beg + SUB);
    for (uint32_t p = beg + threadIdx.x; p < end; p += 256) atomicAdd(&hist[(inter[p] >> g.ib) & (keys - 1)], 1u);
    __syncthreads();
    if (threadIdx.x < keys) {
        const uint32_t cnt = hist[threadIdx.x];
        if (cnt) atomicAdd(&totals_all[(size_t)col * g.nb + bin * keys + threadIdx.x], cnt);
    }
}

// The bin-local scans of the per-bucket totals (the first `keys` lanes of the workgroup, a lane per bucket of the bin; every lane
// of the workgroup must call): the bucket's place in the column's dense entry list (returned) and its distance from the previous
// non-empty bucket of the bin minus one (`esc` for a bin's first one or a longer gap: what the bucket's first entry tells the
// accumulation lane that runs into it).  `publish`: also write what the LATER kernels read — bstart[b], lane_b[] (the bucket an
// accumulation lane starts in), pstart[b] / pbucket[] (the bucket's parts of at most WCAP slots, T1) and the no-part markers at
// the end of the bin's part region.  (Round 3 ran this as a kernel of its own between the fine histogram and the second scatter;
// now every chunk of a bin redoes the two 128-element scans — a few microseconds — and the bin's first chunk publishes.)
struct WideBinScan {
    uint32_t wsum[4];
    int wlast[2];
};
__device__ __forceinline__ uint32_t wide_binscan(WideBinScan& B, const uint32_t* __restrict__ chdr, const WideGeo& g, uint32_t col,
                                                 uint32_t bin, const uint32_t* __restrict__ totals_all, bool publish,
                                                 uint32_t* __restrict__ bstart_all, uint32_t* __restrict__ lane_b,
                                                 uint32_t* __restrict__ pstart_all, uint32_t* __restrict__ pbucket, uint32_t WCAP,
                                                 uint32_t* delta_out) {
    constexpr uint32_t CB = WCB + 1;
    const uint32_t nb = g.nb, keys = 1u << g.fb;
    const uint32_t esc = (1u << (30 - g.ib)) - 1;
    const bool mine = threadIdx.x < keys;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t top = min(63u, keys - 1);  // the last lane of a wave that holds buckets
    const uint32_t b = bin * keys + threadIdx.x;
    const uint32_t cnt = mine ? totals_all[(size_t)col * nb + b] : 0;
    if (threadIdx.x < 4) B.wsum[threadIdx.x] = 0;
    if (threadIdx.x < 2) B.wlast[threadIdx.x] = -1;
    __syncthreads();
    // exclusive scan of the counts -> place in the entry list
    uint32_t xe = cnt;
    int last = cnt ? (int)threadIdx.x : -1;  // inclusive running maximum: the last non-empty bucket up to this one
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t ye = __shfl_up(xe, off);
        const int yl = __shfl_up(last, off);
        if ((int)lane >= off) {
            xe += ye;
            last = max(last, yl);
        }
    }
    if (mine && lane == top) {
        B.wsum[2 * wave] = xe;
        B.wlast[wave] = last;
    }
    __syncthreads();
    const uint32_t s = chdr[bin] + xe - cnt + (wave == 1 ? B.wsum[0] : 0);
    // the last non-empty bucket strictly before this one
    int prev = __shfl_up(last, 1);
    if (lane == 0) prev = -1;
    if (wave == 1) prev = max(prev, B.wlast[0]);
    const uint32_t gap = prev < 0 ? esc : (uint32_t)((int)threadIdx.x - prev - 1);
    *delta_out = gap < esc ? gap : esc;
    if (!publish) return s;  // (workgroup-uniform)
    const uint32_t slots = wide_slot_count(s, cnt);
    const uint32_t np = mine ? (slots + WCAP - 1) / WCAP : 0;
    // exclusive scan of the part counts -> place in the bin's part region
    uint32_t xp = np;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t yp = __shfl_up(xp, off);
        if ((int)lane >= off) xp += yp;
    }
    if (mine && lane == top) B.wsum[2 * wave + 1] = xp;
    __syncthreads();
    const uint32_t pbase = chdr[4 * CB + bin], pend = chdr[4 * CB + bin + 1];
    const uint32_t p_used = B.wsum[1] + B.wsum[3];
    const uint32_t p0 = pbase + xp - np + (wave == 1 ? B.wsum[1] : 0);
    if (mine) {
        bstart_all[(size_t)col * nb + b] = s;
        pstart_all[(size_t)col * nb + b] = p0;
        if (cnt)  // lanes whose first entry lies in this bucket
            for (uint32_t t = (s + WL - 1) / WL; t * WL < s + cnt; t++) lane_b[t] = b;
        for (uint32_t q = 0; q < np; q++) pbucket[p0 + q] = b;
    }
    for (uint32_t q = pbase + p_used + threadIdx.x; q < pend; q += 256) pbucket[q] = 0xffffffffu;
    return s;
}

// level 2 of the sort: one chunk (<= SUB entries) of one coarse bin into its buckets, dense (no padding): the first entry of
// every bucket carries WIDE_FLAG and the bucket's distance from the previous non-empty one in its spare bits.  The bin-local
// scans are the kernel's prologue (wide_binscan); the first chunk of a bin publishes them, one block beyond the worst case of
// chunks covers the bins without entries.
__global__ __launch_bounds__(256) void msm_wscatter2_kernel(const uint32_t* __restrict__ inter_all, size_t inter_stride,
                                                            const uint32_t* __restrict__ coarse_all, uint32_t coarse_stride, WideGeo g,
                                                            const uint32_t* __restrict__ totals_all, uint32_t* __restrict__ bstart_all,
                                                            uint32_t* __restrict__ cursor_all, uint32_t* __restrict__ entries_all,
                                                            size_t ent_stride, uint32_t* __restrict__ lane_b_all, uint32_t lane_stride,
                                                            uint32_t* __restrict__ pstart_all, uint32_t* __restrict__ pbucket_all,
                                                            uint32_t part_stride, uint32_t WCAP) {
    __shared__ SortLds S;
    __shared__ WideBinScan B;
    __shared__ uint32_t s_bin, s_chunk;
    __shared__ uint32_t s_mark[128];  // flag bits of a key's first entry when this chunk holds the bucket's first
    __shared__ uint32_t s_start[128], s_delta[128];
    constexpr uint32_t CB = WCB + 1;
    const uint32_t col = blockIdx.y, nb = g.nb, keys = 1u << g.fb;
    const uint32_t* __restrict__ inter = inter_all + (size_t)col * inter_stride;
    const uint32_t* __restrict__ chdr = coarse_all + (size_t)col * coarse_stride;
    uint32_t* __restrict__ cursor = cursor_all + (size_t)col * nb;
    uint32_t* __restrict__ entries = entries_all + (size_t)col * ent_stride;
    const uint32_t* cpre = chdr + CB;
    if (blockIdx.x == gridDim.x - 1) {
        // the spare block: bins without entries have no chunk, but their buckets' starts are read all the same (an empty bucket
        // shares its start with the next non-empty one: wide_bucket_at) and their part ranges must be empty
        for (uint32_t bin = 0; bin < g.bins; bin++) {
            if (chdr[bin + 1] != chdr[bin]) continue;
            if (threadIdx.x < keys) {
                bstart_all[(size_t)col * nb + bin * keys + threadIdx.x] = chdr[bin];
                pstart_all[(size_t)col * nb + bin * keys + threadIdx.x] = chdr[4 * CB + bin];
            }
        }
        return;
    }
    if (blockIdx.x >= cpre[g.bins]) return;  // the grid is sized for the worst case
    if (threadIdx.x == 0) {
        uint32_t lo = 0, hi = g.bins;  // the bin whose chunk range holds blockIdx.x
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (cpre[mid] <= blockIdx.x) lo = mid;
            else hi = mid;
        }
        s_bin = lo;
        s_chunk = blockIdx.x - cpre[lo];
    }
    if (threadIdx.x < keys) S.cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t bin = s_bin;
    {
        uint32_t dl;
        const uint32_t st = wide_binscan(B, chdr, g, col, bin, totals_all, s_chunk == 0, bstart_all, lane_b_all + (size_t)col * lane_stride,
                                         pstart_all, pbucket_all + (size_t)col * part_stride, WCAP, &dl);
        if (threadIdx.x < keys) {
            s_start[threadIdx.x] = st;
            s_delta[threadIdx.x] = dl;
        }
    }
    const uint32_t beg = chdr[bin] + s_chunk * SUB;
    const uint32_t end = min(chdr[bin + 1], beg + SUB);
    constexpr uint32_t PER = SUB / 256;
    uint32_t ent[PER], meta[PER];
#pragma unroll
    for (uint32_t q = 0; q < PER; q++) {
        const uint32_t p = beg + threadIdx.x + q * 256;
        meta[q] = 0xffffffffu;
        if (p < end) {
            const uint32_t e = inter[p];
            const uint32_t key = (e >> g.ib) & (keys - 1);
            ent[q] = e & ~((keys - 1) << g.ib);
            meta[q] = (key << 16) | atomicAdd(&S.cnt[key], 1u);
        }
    }
    __syncthreads();
    sort_scan(S, keys);
    if (threadIdx.x < keys) {
        const uint32_t cnt = S.cnt[threadIdx.x], b = bin * keys + threadIdx.x;
        const uint32_t before = cnt ? atomicAdd(&cursor[b], cnt) : 0;
        S.gbase[threadIdx.x] = s_start[threadIdx.x] + before;
        s_mark[threadIdx.x] = (cnt && before == 0) ? (WIDE_FLAG | (s_delta[threadIdx.x] << g.ib)) : 0;
    }
    __syncthreads();
#pragma unroll
    for (uint32_t q = 0; q < PER; q++)
        if (meta[q] != 0xffffffffu) {
            const uint32_t key = meta[q] >> 16, pos = S.lstart[key] + (meta[q] & 0xffffu);
            S.sorted[pos] = ent[q];
            S.kid[pos] = (uint8_t)key;
        }
    __syncthreads();
    const uint32_t total = end - beg;
    for (uint32_t q = threadIdx.x; q < total; q += 256) {
        const uint32_t key = S.kid[q];
        entries[S.gbase[key] + q - S.lstart[key]] = S.sorted[q] | (q == S.lstart[key] ? s_mark[key] : 0u);
    }
}

// the counters a pass of the wide path expects to be zero: per-bucket totals and level-2 cursors, the append cursors of the
// coarse bins, counts[].  Launched only when the workspace is not known to be clean (first pass, after an error or after the
// 13-bit plan used the workspace): every pass leaves them clean (H1 zeroes the other set of per-bucket counters and the redo
// count, the fine histogram's spare block the append cursors)
__global__ void msm_wclear_kernel(uint32_t* __restrict__ totals, uint32_t* __restrict__ cursor, uint32_t* __restrict__ totals2,
                                  uint32_t* __restrict__ cursor2, uint32_t nbt, uint32_t* __restrict__ counts,
                                  uint32_t* __restrict__ coarse, uint32_t coarse_stride, uint32_t ncols) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nbt) {
        totals[i] = 0;
        cursor[i] = 0;
        totals2[i] = 0;
        cursor2[i] = 0;
    }
    if (i < 4 * (MSM_MAX_BATCH + 1)) counts[i] = 0;
    if (i < ncols * WCB) coarse[(size_t)(i / WCB) * coarse_stride + WCUR0 + (i % WCB) * WCUR] = 0;
}

// the bucket of the entry at position `pos` of a column (the escape of a first-of-bucket entry whose distance field is
// saturated): the last bucket whose start is <= pos — empty buckets share their start with the next non-empty one
__device__ __noinline__ uint32_t wide_bucket_at(const uint32_t* __restrict__ bstart, uint32_t nb, uint32_t pos) {
    uint32_t lo = 0, hi = nb;  // bstart[lo] <= pos (bstart[0] = 0), hi: first index with bstart > pos (or nb)
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (bstart[mid] <= pos) lo = mid;
        else hi = mid;
    }
    return lo;
}

// ---- the accumulation of the wide path: lane t of a column sums entries [t WL, (t + 1) WL) of the dense list, storing its
// running sum and restarting at every first-of-bucket entry (slot index = t + bucket).  SAFE as in accumulate_segment:
// the unchecked loop vouches that the table holds no identity and reports a lane whose sums show an exceptional step
// (ZZ = 0) to the redo list; the checked loop is exact for any table.
template <bool SAFE>
__device__ __forceinline__ bool wide_accumulate_lane(const uint32_t* __restrict__ e, uint32_t count, uint32_t pos0, uint32_t t, uint32_t b,
                                                     const uint32_t* __restrict__ bstart, uint32_t nb,
                                                     const G1Affine* __restrict__ table, G1X29S* __restrict__ slots, uint32_t ib) {
    G1X29 acc;
    acc.inf = true;
    bool suspicious = false;
    const uint32_t idx_mask = (1u << ib) - 1, esc = (1u << (30 - ib)) - 1;
    for (uint32_t k = 0; k < count; k++) {
        const uint32_t y = e[k];
        if ((y & WIDE_FLAG) && k) {  // a new bucket begins inside the lane's run
            if (!SAFE && !acc.inf && is_zero29(acc.zz)) suspicious = true;
            g1x29_store(slots + t + b, acc);
            acc.inf = true;
            const uint32_t d = (y >> ib) & esc;
            b = d < esc ? b + 1 + d : wide_bucket_at(bstart, nb, pos0 + k);
        }
        G1Affine p = affine_load(table + (y & idx_mask));
        if (SAFE && affine_is_identity(p)) continue;
        if (y & SIGN_BIT) p.y = fe_neg(p.y);
        if (!g1x29_add_affine<SAFE, true>(acc, p.x, p.y)) {
            // same x as the running sum (doubling or cancellation): the general formulas, rarely.  The table is in the
            // internal form: back to the standard one for the general addition (divide by 32: one product each)
            G1X s = g1x29_to_std(acc);
            Fq px = internal_to_std(to29(p.x)), py = internal_to_std(to29(p.y));
            g1x_add_affine(s, px, py);
            acc = g1x29_from_std(s);
        }
    }
    if (!SAFE && !acc.inf && is_zero29(acc.zz)) suspicious = true;
    g1x29_store(slots + t + b, acc);
    return suspicious;
}

#if ZK_ACC_WAVES
__attribute__((amdgpu_waves_per_eu(ZK_ACC_WAVES, ZK_ACC_WAVES)))
#endif
__global__ __launch_bounds__(64) void msm_wacc_kernel(const uint32_t* __restrict__ entries_all, size_t ent_stride,
                                                      const G1Affine* __restrict__ table, const uint32_t* __restrict__ counts,
                                                      const uint32_t* __restrict__ lane_b_all, uint32_t lane_stride,
                                                      const uint32_t* __restrict__ bstart_all, uint32_t nb,
                                                      G1X29S* __restrict__ slot_all, uint32_t slot_stride, uint32_t ib) {
    const uint32_t col = blockIdx.y, total = counts[4 * col];
    const uint32_t t = blockIdx.x * 64 + threadIdx.x;
    if (t * WL >= total) return;
    wide_accumulate_lane<true>(entries_all + (size_t)col * ent_stride + (size_t)t * WL, min(WL, total - t * WL), t * WL, t,
                               lane_b_all[(size_t)col * lane_stride + t], bstart_all + (size_t)col * nb, nb, table,
                               slot_all + (size_t)col * slot_stride, ib);
}
#if ZK_ACC_WAVES
__attribute__((amdgpu_waves_per_eu(ZK_ACC_WAVES, ZK_ACC_WAVES)))
#endif
__global__ __launch_bounds__(64) void msm_wacc_fast_kernel(const uint32_t* __restrict__ entries_all, size_t ent_stride,
                                                           const G1Affine* __restrict__ table, uint32_t* __restrict__ counts,
                                                           const uint32_t* __restrict__ lane_b_all, uint32_t lane_stride,
                                                           const uint32_t* __restrict__ bstart_all, uint32_t nb,
                                                           G1X29S* __restrict__ slot_all, uint32_t slot_stride, uint32_t* __restrict__ redo,
                                                           uint32_t ib) {
    const uint32_t col = blockIdx.y, total = counts[4 * col];
    const uint32_t t = blockIdx.x * 64 + threadIdx.x;
    if (t * WL >= total) return;
    if (wide_accumulate_lane<false>(entries_all + (size_t)col * ent_stride + (size_t)t * WL, min(WL, total - t * WL), t * WL, t,
                                    lane_b_all[(size_t)col * lane_stride + t], bstart_all + (size_t)col * nb, nb, table,
                                    slot_all + (size_t)col * slot_stride, ib))
        redo[atomicAdd(&counts[1], 1u)] = col * lane_stride + t;  // at most one entry per lane: redo[] has one word each
}
__global__ __launch_bounds__(64) void msm_wacc_redo_kernel(const uint32_t* __restrict__ entries_all, size_t ent_stride,
                                                           const G1Affine* __restrict__ table, const uint32_t* __restrict__ counts,
                                                           const uint32_t* __restrict__ lane_b_all, uint32_t lane_stride,
                                                           const uint32_t* __restrict__ bstart_all, uint32_t nb,
                                                           G1X29S* __restrict__ slot_all, uint32_t slot_stride, const uint32_t* __restrict__ redo,
                                                           uint32_t ib) {
    const uint32_t m = counts[1];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
        const uint32_t col = redo[i] / lane_stride, t = redo[i] - col * lane_stride, total = counts[4 * col];
        wide_accumulate_lane<true>(entries_all + (size_t)col * ent_stride + (size_t)t * WL, min(WL, total - t * WL), t * WL, t,
                                   lane_b_all[(size_t)col * lane_stride + t], bstart_all + (size_t)col * nb, nb, table,
                                   slot_all + (size_t)col * slot_stride, ib);
    }
}

// T1: part g of a column = up to WCAP consecutive slots of one bucket, summed serially by one lane; then the lanes of a wave
// that hold parts of the same bucket (a bucket's parts are consecutive) are joined by a segmented shuffle tree — the first
// lane of every run stores.  part[g] is therefore valid at the "heads": g = pstart[b] and the multiples of 64 inside
// (pstart[b], pstart[b + 1]).
__global__ __launch_bounds__(64) void msm_wparts_kernel(const G1X29S* __restrict__ slot_all, uint32_t slot_stride,
                                                        const uint32_t* __restrict__ totals_all, const uint32_t* __restrict__ bstart_all,
                                                        const uint32_t* __restrict__ pstart_all, const uint32_t* __restrict__ pbucket_all,
                                                        uint32_t part_stride, uint32_t nb, const uint32_t* __restrict__ counts,
                                                        G1X29S* __restrict__ part_all, uint32_t WCAP, uint32_t lmin) {
    const uint32_t col = blockIdx.y;
    const uint32_t nparts = counts[4 * col + 2];
    if (blockIdx.x * 64 >= nparts) return;  // wave-uniform
    const uint32_t lane = threadIdx.x;
    const uint32_t g = blockIdx.x * 64 + lane;
    const G1X29S* __restrict__ slots = slot_all + (size_t)col * slot_stride;
    uint32_t b = g < nparts ? pbucket_all[(size_t)col * part_stride + g] : 0xffffffffu;  // 0xffffffff: the unused end of a bin's part region
    bool active = b != 0xffffffffu;
    uint32_t s = 0, s_end = 0;
    if (active) {
        // the bucket's slots: one per accumulation lane that held some of its entries, at index lane + bucket
        const uint32_t e0 = bstart_all[(size_t)col * nb + b];
        const uint32_t s0 = e0 / WL + b;
        const uint32_t len = wide_slot_count(e0, totals_all[(size_t)col * nb + b]);
        if (len <= lmin) {  // the per-bucket kernel (msm_wbucket_kernel) has summed this bucket: its parts take no lane here
            active = false;
            b = 0xfffffffeu - lane;  // (distinct: no two such lanes look like parts of one bucket)
        }
    }
    if (!__any(active)) return;  // wave-uniform: under lmin > 0 nearly every wave of a uniformly random column
    if (active) {
        const uint32_t e0 = bstart_all[(size_t)col * nb + b];
        const uint32_t s0 = e0 / WL + b;
        const uint32_t len = wide_slot_count(e0, totals_all[(size_t)col * nb + b]);
        const uint32_t np = (len + WCAP - 1) / WCAP;
        const uint32_t p = g - pstart_all[(size_t)col * nb + b];
        // balanced shares: part p of np takes slots [s0 + p len / np, s0 + (p + 1) len / np)
        s = s0 + (uint32_t)(((uint64_t)p * len) / np);
        s_end = s0 + (uint32_t)(((uint64_t)(p + 1) * len) / np);
    }
    G1X29 acc = g1x29_identity();
    int off = 1;
#pragma unroll 1
    for (;;) {
        G1X29 v;
        bool have;
        if (__any(s < s_end)) {  // wave-uniform: the serial runs
            have = s < s_end;
            if (have) v = g1x29_load(slots + s);
            s++;
        } else {
            // segmented tree: lane i takes lane i + off's sum when both hold parts of the same bucket (a bucket's parts are
            // consecutive lanes: a level without any such pair ends the tree)
            if (off >= 64) break;
            const uint32_t kb = (uint32_t)__shfl_down((int)b, off);
            have = active && lane + off < 64 && kb == b;
            if (!__any(have)) break;
            v = g1x29_shfl_down(acc, off);
            off <<= 1;
        }
        if (have) g1x29_add<ZK_TAIL_SER>(acc, v);
    }
    const uint32_t prev = (uint32_t)__shfl_up((int)b, 1);
    if (active && (lane == 0 || prev != b)) g1x29_store(part_all + (size_t)col * part_stride + g, acc);
}

// T1, per-bucket form (round 6): ONE lane per bucket sums all of the bucket's slots serially — no parts, no shuffle tree.  A wave
// of the part form spends 7.9 addition times on 64 parts = ~21 buckets (5 .. 6 serial additions + two tree levels in which most
// lanes idle: 4.35e7 instructions per 2^19 column, profiles/r5_pmc_ops.txt); here a wave takes 64 buckets in max(len) - 1
// additions (len = 17 .. 19 slots for a uniformly random column) with every lane busy, and the products take the serial
// multiply-add form: fewer instructions, a LONGER dependent chain — the form for a loaded chip (the pass's tail on the main
// stream: three or more proofs in flight), while a lone proof keeps the part form, whose chain is half as long.  Buckets of
// more than `lmax` slots (witness-like columns put 400 K entries into a few buckets) are left to msm_wparts_kernel (lmin = lmax).
// The sum goes where T2 looks for it: part[pstart[b]]; a further head of the bucket's part range (a multiple of 64 inside it)
// becomes the identity.
__global__ __launch_bounds__(64) void msm_wbucket_kernel(const G1X29S* __restrict__ slot_all, uint32_t slot_stride,
                                                         const uint32_t* __restrict__ totals_all, const uint32_t* __restrict__ bstart_all,
                                                         const uint32_t* __restrict__ pstart_all, uint32_t part_stride, uint32_t nb,
                                                         G1X29S* __restrict__ part_all, uint32_t WCAP, uint32_t lmax) {
    const uint32_t col = blockIdx.y, b = blockIdx.x * 64 + threadIdx.x;  // nb is a multiple of 64
    const G1X29S* __restrict__ slots = slot_all + (size_t)col * slot_stride;
    const uint32_t e0 = bstart_all[(size_t)col * nb + b];
    const uint32_t len = wide_slot_count(e0, totals_all[(size_t)col * nb + b]);
    const bool mine = len != 0 && len <= lmax;
    uint32_t s = e0 / WL + b;
    const uint32_t s_end = mine ? s + len : s;
    G1X29 acc = g1x29_identity();
#pragma unroll 1
    while (__any(s < s_end)) {  // wave-uniform
        const bool have = s < s_end;
        G1X29 v;
        if (have) v = g1x29_load(slots + s);
        s++;
        if (have) g1x29_add<ZK_T1B_SER>(acc, v);
    }
    if (mine) {
        G1X29S* __restrict__ part = part_all + (size_t)col * part_stride;
        const uint32_t g = pstart_all[(size_t)col * nb + b], g_end = g + (len + WCAP - 1) / WCAP;
        g1x29_store(part + g, acc);
        const uint32_t h = (g | 63u) + 1;
        if (h < g_end) g1x29_store(part + h, g1x29_identity());
    }
}

// T2: one wave per row (blockIdx.x < rows) or column (blockIdx.x - rows) of the column's bucket matrix [rows][256]: lanes walk
// their buckets' heads serially, then a shuffle tree.  rc[col][rows + 256]
__global__ __launch_bounds__(64) void msm_wrowcol_kernel(const G1X29S* __restrict__ part_all, uint32_t part_stride,
                                                         const uint32_t* __restrict__ totals_all, const uint32_t* __restrict__ bstart_all,
                                                         const uint32_t* __restrict__ pstart_all, uint32_t nb,
                                                         G1X29S* __restrict__ rc_all, uint32_t WCAP) {
    const uint32_t col = blockIdx.y, rows = nb >> 8, r = blockIdx.x, lane = threadIdx.x;
    const uint32_t* __restrict__ pstart = pstart_all + (size_t)col * nb;
    const uint32_t* __restrict__ totals = totals_all + (size_t)col * nb;
    const uint32_t* __restrict__ bstart = bstart_all + (size_t)col * nb;
    const G1X29S* __restrict__ part = part_all + (size_t)col * part_stride;
    const bool is_row = r < rows;
    // lane's j-th bucket: rows: 256 r + lane + 64 j (j < 4); columns: 256 (lane + 64 j) + (r - rows) (j < rows / 64)
    const uint32_t nj = is_row ? 4u : rows / 64;
    uint32_t j = 0, g = 0, g_end = 0;
    const auto bucket_of = [&](uint32_t jj) { return is_row ? 256 * r + lane + 64 * jj : 256 * (lane + 64 * jj) + (r - rows); };
    const auto open_bucket = [&]() {
        while (j < nj) {
            const uint32_t b = bucket_of(j);
            g = pstart[b];
            g_end = g + (wide_slot_count(bstart[b], totals[b]) + WCAP - 1) / WCAP;
            if (g < g_end) return;
            j++;
        }
    };
    open_bucket();
    G1X29 acc = g1x29_identity();
    int off = 32;
#pragma unroll 1
    for (;;) {
        G1X29 v;
        bool have;
        if (__any(j < nj)) {  // wave-uniform
            have = j < nj;
            if (have) {
                v = g1x29_load(part + g);
                g = (g | 63u) + 1;  // the next head of this bucket, if any
                if (g >= g_end) {
                    j++;
                    open_bucket();
                }
            }
        } else {
            if (off == 0) break;
            v = g1x29_shfl_down(acc, off);
            have = (int)lane < off;
            off >>= 1;
        }
        if (have) g1x29_add<ZK_TAIL_SER>(acc, v);
    }
    if (lane == 0) g1x29_store(rc_all + (size_t)col * (rows + 256) + r, acc);
}

// T3: blockIdx.x = t < 9: sum of the column sums C_l with bit t of (l + 1) set; t >= 9: sum of the row sums R_h with bit
// t - 9 of h set.  One wave each; lane 0 hands the sum over in the standard form.  out[col][WIDE_SUMS]
__global__ __launch_bounds__(64) void msm_wbits_kernel(const G1X29S* __restrict__ rc_all, uint32_t nb, G1X* __restrict__ out,
                                                       const uint32_t* __restrict__ counts) {
    const uint32_t col = blockIdx.y, rows = nb >> 8, t = blockIdx.x, lane = threadIdx.x;
    // the word after the last column's sums tells the host how many lanes of the unchecked accumulation saw an exceptional
    // step (same x: possible only over a degenerate basis): it then re-runs those lanes and this tail (msm_wide_redo) — the
    // common case pays no redo launch
    if (col == 0 && t == 0 && lane == 0) *reinterpret_cast<uint32_t*>(out + (size_t)gridDim.y * WIDE_SUMS) = counts[1];
    const G1X29S* __restrict__ rc = rc_all + (size_t)col * (rows + 256);
    const bool cols = t < 9;
    const uint32_t items = cols ? 256u : rows;
    uint32_t i = lane;
    const auto wanted = [&](uint32_t ii) { return cols ? (((ii + 1) >> t) & 1u) != 0 : ((ii >> (t - 9)) & 1u) != 0; };
    while (i < items && !wanted(i)) i += 64;
    G1X29 acc = g1x29_identity();
    int off = 32;
#pragma unroll 1
    for (;;) {
        G1X29 v;
        bool have;
        if (__any(i < items)) {  // wave-uniform
            have = i < items;
            if (have) {
                v = g1x29_load(rc + (cols ? rows + i : i));
                i += 64;
                while (i < items && !wanted(i)) i += 64;
            }
        } else {
            if (off == 0) break;
            v = g1x29_shfl_down(acc, off);
            have = (int)lane < off;
            off >>= 1;
        }
        if (have) g1x29_add<ZK_TAIL_SER>(acc, v);
    }
    if (lane == 0) {
        G1X r = G1X::identity();
        if (!acc.inf) {
            r.x = internal_to_std_call(acc.x);
            r.y = internal_to_std_call(acc.y);
            r.zz = internal_to_std_call(acc.zz);
            r.zzz = internal_to_std_call(acc.zzz);
        }
        g1x_store(out + (size_t)col * WIDE_SUMS + t, r);
    }
}

// ------------------------------------------------------ fixed-base tables ---
// table[w][i] = 2^(c w) * P_i (affine).  One launch per window: c doublings + one inversion.
__global__ __launch_bounds__(64) void msm_table_step_kernel(const G1Affine* __restrict__ prev, G1Affine* __restrict__ next,
                                                            uint32_t n, uint32_t c) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const G1Affine p = affine_load(prev + i);
    G1Affine r;
    if (affine_is_identity(p)) {
        r.x = Fq::zero();
        r.y = Fq::zero();
    } else {
        G1X acc = g1x_dbl_affine(p.x, p.y);
        for (uint32_t k = 1; k < c; k++) acc = g1x_dbl(acc);
        if (acc.is_identity()) {  // cannot happen on a prime-order curve; kept for completeness
            r.x = Fq::zero();
            r.y = Fq::zero();
        } else {
            const Fq t = fe_inv(acc.zzz);
            const Fq u = fe_mul(acc.zz, t);
            r.x = fe_mul(acc.x, fe_sqr(u));
            r.y = fe_mul(acc.y, t);
        }
    }
    fe_store(&next[i].x, r.x);
    fe_store(&next[i].y, r.y);
}

// does any of the n points equal the identity (0, 0)?  Decides which accumulation loop a basis gets (msm_accumulate_kernel).
__global__ void msm_identity_flag_kernel(const G1Affine* __restrict__ b, uint32_t n, uint32_t* __restrict__ flag) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && affine_is_identity(affine_load(b + i))) atomicOr(flag, 1u);
}
hipError_t msm_bases_have_identity(const G1Affine* bases, uint32_t n, hipStream_t st, uint32_t* d_word, uint32_t* h_word, bool* out) {
    hipError_t e = hipMemsetAsync(d_word, 0, 4, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(msm_identity_flag_kernel, dim3((n + 255) / 256), dim3(256), 0, st, bases, n, d_word);
    if ((e = hipMemcpyAsync(h_word, d_word, 4, hipMemcpyDeviceToHost, st)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
    *out = *h_word != 0;
    return hipSuccess;
}

// x * 2^256 (standard memory form) -> x * 2^261 (the accumulation's internal form, canonical words): times 32
__global__ void msm_table_internal_kernel(G1Affine* __restrict__ t, size_t count) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    G1Affine p = affine_load(t + i);
    for (int k = 0; k < 5; k++) {
        p.x = fe_add(p.x, p.x);
        p.y = fe_add(p.y, p.y);
    }
    fe_store(&t[i].x, p.x);
    fe_store(&t[i].y, p.y);
}

// the same rule msm_run applies (wide workspace && table stride == workspace length; workspaces are >= 1024 long)
bool msm_table_is_internal(uint32_t c, size_t n) { return msm_wide_applies(c, n) && n >= 1024; }

hipError_t msm_build_table(const G1Affine* bases, uint32_t n, uint32_t c, G1Affine* table, hipStream_t st) {
    const uint32_t nwin = nwin_for(c);
    hipError_t e = hipMemcpyAsync(table, bases, (size_t)n * sizeof(G1Affine), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return e;
    for (uint32_t w = 1; w < nwin; w++)
        hipLaunchKernelGGL(msm_table_step_kernel, dim3((n + 63) / 64), dim3(64), 0, st, table + (size_t)(w - 1) * n,
                           table + (size_t)w * n, n, c);
    if (msm_table_is_internal(c, n)) {
        // the wide path reads its window tables in the accumulation's internal form (the identity stays (0, 0))
        const size_t count = (size_t)nwin * n;
        hipLaunchKernelGGL(msm_table_internal_kernel, dim3((uint32_t)((count + 255) / 256)), dim3(256), 0, st, table, count);
    }
    return hipGetLastError();
}

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
        // T1 per bucket where the tail shares its stream with the head (the engine puts it there while three or more proofs are in
        // flight on the device: issue slots are what is short), per part — the shorter dependent chain — for a lone proof
        const bool per_bucket = ws->w_t1_mode == 1 || (ws->w_t1_mode == 0 && ts == st);
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
