// msm_wide.hip.h — the wide path: 15 / 16 / 17-bit windows over the resident SRS's window tables, ONE bucket set per column
// (part of msm.hip's translation unit, inside namespace zk).  Kernels: the head (msm_whist, msm_wscatter1, msm_wfinehist,
// msm_wscatter2: digits recomputed from the scalars, two-level counting sort into a dense entry list), the accumulation
// (msm_wacc_fast / msm_wacc / msm_wacc_redo: restartable lanes of 16 entries, one partial sum per lane and bucket) and the
// reduction tail (T1 msm_wparts / msm_wbucket, T2 msm_wrowcol, T3 msm_wbits).  Host side: msm_run_wide in msm.hip.
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

// T1, per-bucket form (round 6, ZK_OPT_MSM_T1 = 1; NOT the default): ONE lane per bucket sums all of the bucket's slots serially —
// no parts, no shuffle tree.  A wave of the part form spends 7.9 addition times on 64 parts = ~21 buckets (5 .. 6 serial additions
// + two tree levels in which most lanes idle: 4.35e7 instructions per 2^19 column, profiles/r5_pmc_ops.txt); here a wave takes
// 64 buckets in max(len) - 1 additions (len = 17 .. 19 slots for a uniformly random column) with every lane busy, and the
// products take the serial multiply-add form: a quarter fewer instructions, a dependent chain twice as long.  Measured
// (profiles/r6_ab_t1_per_bucket.txt, one box): the tail of a lone 2^19 column 0.304 -> 0.339 ms, of a two-column pass 0.427 ->
// 0.377 ms; whole proofs under four pipelines 98.8 / 99.9 (parts) against 99.7 / 99.4 proofs/s, a lone proof 11.41 / 11.42
// against 11.52 / 11.46 ms: nothing for throughput — the chip is not short of issue slots where T1 runs — and slower alone, so
// the part form stays the default.  Buckets of more than `lmax` slots (witness-like columns put 400 K entries into a few
// buckets) are left to msm_wparts_kernel (lmin = lmax).  The sum goes where T2 looks for it: part[pstart[b]]; a further head
// of the bucket's part range (a multiple of 64 inside it) becomes the identity.
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

