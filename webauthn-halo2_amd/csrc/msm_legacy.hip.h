// msm_legacy.hip.h — the window plans of the SMALL and the VERY LARGE fixed-base MSMs and of arbitrary bases (part of msm.hip's
// translation unit, inside namespace zk): signed-digit recoding into int16 planes, the one-pass and the two-level counting
// sort, the padded 16-entry accumulation segments, the two-level gather and the bit sums.  Who takes it: n < 2^16 over the
// resident SRS (lg n - 5 window bits: a bucket set of its own per column is too much tail for so few points — tools/bench_rows.py:
// k = 15 7.0 ms on this plan against 7.6 / 7.9 with 15 / 16-bit windows on the wide path), n >= 2^22 (table indexes beyond
// 26 bits: 15 bits on the swept sort), and zk_msm_bn254 (arbitrary bases: W windows x 2^(c-1) buckets, host Horner).
// The k = 16 .. 21 proofs of the reference's configurations run on msm_wide.hip.h.
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

