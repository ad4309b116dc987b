// msm.hip — Pippenger bucket multi-scalar multiplication over BN254 G1 for gfx950.
//
// Device replacement for halo2_proofs `arithmetic::best_multiexp`, reached only
// through `ParamsKZG::{commit, commit_lagrange}` (reference
// call sites halo2-circuits/src/ecc/ecdsa_p256.rs:366-373,416-423,555-562).
// Any correct algorithm yields the same group element, so the result is
// bit-identical to the reference's after affine normalisation.
//
// Pipeline (all on one stream, no host round trip until the W window sums):
//   1. msm_digits<COUNT>   signed c-bit window recoding of every scalar, bucket histogram
//   2. msm_scan            exclusive scan of the histogram -> bucket cursors
//   3. msm_digits<SCATTER> counting-sort scatter of (bucket, +-point index) entries
//   4. msm_accumulate      fixed-length segments of the sorted entry list, one
//                          thread each, mixed XYZZ adds; runs that lie inside a
//                          segment go straight to their bucket, the first/last
//                          run of a segment becomes a "slot"
//   5. msm_reduce_slots    the same segmented reduction over the slot list until
//                          it is short, then msm_finalize_slots
//   6. msm_bucket_reduce   bucket j -> (j+1) * B_j by double-and-add, LDS tree sum
//   7. msm_window_reduce   per-window sum of the block partials
//   8. host                Horner over the W window sums (W * c doublings; serial
//                          work that a single GPU lane would take ~ms to do)
// Load balance does not depend on the scalar distribution: witness columns are
// dominated by zeros / small values (hot low buckets), and a segment is a fixed
// number of entries whatever bucket they fall in.
#include <string.h>
#include <vector>

#include "engine.h"

namespace zk {

static constexpr uint32_t SIGN_BIT = 0x80000000u;

struct MsmWorkspace {
    size_t max_n;
    uint32_t c, nwin, nb;       // window bits, windows, buckets per window
    uint32_t seg0, seg1;        // entries per thread at level 0 / slot levels
    uint32_t* hist;             // [nwin*nb + 1]
    uint32_t* cursor;           // [nwin*nb + 1]
    uint32_t* counts;           // [16] device-side list lengths per level (counts[0] = #entries)
    uint2* entries;             // [max_n * nwin]
    uint32_t* slot_bucket[2];
    G1X* slot_pt[2];
    size_t slot_cap;
    G1X* bucket_sum;            // [nwin*nb]
    G1X* block_sum;             // [nwin*nb/256]
    G1X* window_sum;            // [nwin]
};

uint32_t msm_auto_window(size_t n) {
    uint32_t lg = 0;
    while (((size_t)1 << (lg + 1)) <= n) lg++;
    int c = (int)lg - 6;
    if (c < 9) c = 9;
    if (c > 14) c = 14;
    return (uint32_t)c;
}

static inline uint32_t nwin_for(uint32_t c) { return 254 / c + 1; }

size_t msm_ws_max_n(const MsmWorkspace* ws) { return ws->max_n; }

// ---------------------------------------------------------------- digits ---

template <bool SCATTER>
__global__ __launch_bounds__(256) void msm_digits_kernel(const Fr* __restrict__ scalars, uint32_t n, uint32_t c,
                                                         uint32_t nwin, uint32_t* __restrict__ ctr,
                                                         uint2* __restrict__ entries) {
    __shared__ uint32_t limbs[256][9];  // canonical scalar per thread (+1 pad word, also breaks bank stride)
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Fr s = fe_from_mont(fe_load(scalars + i));
    uint32_t* L = limbs[threadIdx.x];
#pragma unroll
    for (int k = 0; k < 8; k++) L[k] = s.v[k];
    L[8] = 0;
    if (s.is_zero()) return;
    const uint32_t nb = 1u << (c - 1);
    const uint32_t mask = (1u << c) - 1;
    uint32_t carry = 0;
    for (uint32_t w = 0; w < nwin; w++) {
        const uint32_t bit = w * c, word = bit >> 5, off = bit & 31;
        uint32_t raw = 0;
        if (word < 8) {
            uint64_t two = (uint64_t)L[word] | ((uint64_t)L[word + 1] << 32);
            raw = (uint32_t)(two >> off) & mask;
        }
        raw += carry;
        uint32_t mag, neg;
        if (raw > nb) {
            mag = (1u << c) - raw;
            neg = SIGN_BIT;
            carry = 1;
        } else {
            mag = raw;
            neg = 0;
            carry = 0;
        }
        if (mag) {
            const uint32_t bucket = w * nb + (mag - 1);
            const uint32_t pos = atomicAdd(&ctr[bucket], 1u);
            if (SCATTER) entries[pos] = make_uint2(bucket, i | neg);
        }
    }
}

// exclusive scan of hist[0..m) into cursor[0..m], cursor[m] = total = counts[0]
__global__ __launch_bounds__(1024) void msm_scan_kernel(const uint32_t* __restrict__ hist, uint32_t* __restrict__ cursor,
                                                        uint32_t m, uint32_t* __restrict__ counts) {
    __shared__ uint32_t part[1024];
    const uint32_t chunk = (m + 1023) / 1024;
    const uint32_t lo = threadIdx.x * chunk;
    const uint32_t hi = min(m, lo + chunk);
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; i++) sum += hist[i];
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
        const uint32_t h = hist[i];
        cursor[i] = run;
        run += h;
    }
    if (threadIdx.x == 1023) {
        cursor[m] = part[1023];
        counts[0] = part[1023];
    }
}

// ------------------------------------------------------------ accumulate ---

// Level 0: entries are (bucket, +-base index).  Emits exactly two slots per
// active thread: (first bucket, first-run sum) and (last bucket, last-run sum or
// identity when the segment is a single run); runs strictly inside the segment
// are complete buckets and are written directly.
__global__ __launch_bounds__(64) void msm_accumulate_kernel(const uint2* __restrict__ entries,
                                                            const G1Affine* __restrict__ bases,
                                                            const uint32_t* __restrict__ counts, uint32_t seg,
                                                            G1X* __restrict__ bucket_sum,
                                                            uint32_t* __restrict__ slot_bucket,
                                                            G1X* __restrict__ slot_pt, uint32_t* __restrict__ counts_out) {
    const uint32_t total = counts[0];
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nthreads = (total + seg - 1) / seg;
    if (t == 0) *counts_out = 2 * nthreads;
    if (t >= nthreads) return;
    const uint32_t beg = t * seg;
    const uint32_t end = min(total, beg + seg);
    const uint32_t first_b = entries[beg].x;
    uint32_t cur = first_b;
    G1X acc = G1X::identity();
    bool first_open = true;
    for (uint32_t pos = beg; pos < end; pos++) {
        const uint2 e = entries[pos];
        if (e.x != cur) {
            if (first_open) {
                slot_bucket[2 * t] = cur;
                g1x_store(slot_pt + 2 * t, acc);
                first_open = false;
            } else {
                g1x_store(bucket_sum + cur, acc);
            }
            acc = G1X::identity();
            cur = e.x;
        }
        G1Affine p = affine_load(bases + (e.y & ~SIGN_BIT));
        if (affine_is_identity(p)) continue;
        if (e.y & SIGN_BIT) p.y = fe_neg(p.y);
        g1x_add_affine(acc, p.x, p.y);
    }
    if (first_open) {  // one run covers the whole segment
        slot_bucket[2 * t] = cur;
        g1x_store(slot_pt + 2 * t, acc);
        acc = G1X::identity();
    }
    slot_bucket[2 * t + 1] = cur;
    g1x_store(slot_pt + 2 * t + 1, acc);
}

// Slot levels: same segmented reduction with XYZZ + XYZZ adds.
__global__ __launch_bounds__(64) void msm_reduce_slots_kernel(const uint32_t* __restrict__ in_bucket,
                                                              const G1X* __restrict__ in_pt,
                                                              const uint32_t* __restrict__ count_in, uint32_t seg,
                                                              G1X* __restrict__ bucket_sum,
                                                              uint32_t* __restrict__ out_bucket,
                                                              G1X* __restrict__ out_pt, uint32_t* __restrict__ count_out) {
    const uint32_t total = *count_in;
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t nthreads = (total + seg - 1) / seg;
    if (t == 0) *count_out = 2 * nthreads;
    if (t >= nthreads) return;
    const uint32_t beg = t * seg;
    const uint32_t end = min(total, beg + seg);
    uint32_t cur = in_bucket[beg];
    G1X acc = G1X::identity();
    bool first_open = true;
    for (uint32_t pos = beg; pos < end; pos++) {
        const uint32_t b = in_bucket[pos];
        if (b != cur) {
            if (first_open) {
                out_bucket[2 * t] = cur;
                g1x_store(out_pt + 2 * t, acc);
                first_open = false;
            } else {
                g1x_store(bucket_sum + cur, acc);
            }
            acc = G1X::identity();
            cur = b;
        }
        const G1X p = g1x_load(in_pt + pos);
        g1x_add(acc, p);
    }
    if (first_open) {
        out_bucket[2 * t] = cur;
        g1x_store(out_pt + 2 * t, acc);
        acc = G1X::identity();
    }
    out_bucket[2 * t + 1] = cur;
    g1x_store(out_pt + 2 * t + 1, acc);
}

// Last level: one thread per run start walks its run serially.
__global__ __launch_bounds__(64) void msm_finalize_slots_kernel(const uint32_t* __restrict__ in_bucket,
                                                                const G1X* __restrict__ in_pt,
                                                                const uint32_t* __restrict__ count_in,
                                                                G1X* __restrict__ bucket_sum) {
    const uint32_t total = *count_in;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const uint32_t b = in_bucket[i];
    if (i > 0 && in_bucket[i - 1] == b) return;
    G1X acc = g1x_load(in_pt + i);
    for (uint32_t k = i + 1; k < total && in_bucket[k] == b; k++) {
        const G1X p = g1x_load(in_pt + k);
        g1x_add(acc, p);
    }
    g1x_store(bucket_sum + b, acc);
}

__global__ void msm_clear_buckets_kernel(G1X* __restrict__ bucket_sum, uint32_t m) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) g1x_store(bucket_sum + i, G1X::identity());
}

// ---------------------------------------------------------------- reduce ---

__device__ __forceinline__ void block_tree_sum(G1X* sh, G1X& mine, uint32_t nthreads) {
    g1x_store(sh + threadIdx.x, mine);
    __syncthreads();
    for (uint32_t s = nthreads >> 1; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            G1X a = g1x_load(sh + threadIdx.x);
            const G1X b = g1x_load(sh + threadIdx.x + s);
            g1x_add(a, b);
            g1x_store(sh + threadIdx.x, a);
        }
        __syncthreads();
    }
    mine = g1x_load(sh);
}

// block_sum[blk] = sum over the block's 256 buckets of (j+1) * B_j
__global__ __launch_bounds__(256) void msm_bucket_reduce_kernel(const G1X* __restrict__ bucket_sum, uint32_t nb,
                                                                G1X* __restrict__ block_sum) {
    __shared__ G1X sh[256];
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    const uint32_t k = (g & (nb - 1)) + 1;  // multiplier
    const G1X p = g1x_load(bucket_sum + g);
    G1X acc = G1X::identity();
    if (!p.is_identity()) {
        acc = p;
        int top = 31 - __clz(k);
        for (int bit = top - 1; bit >= 0; bit--) {
            acc = g1x_dbl(acc);
            if ((k >> bit) & 1) g1x_add(acc, p);
        }
    }
    block_tree_sum(sh, acc, 256);
    if (threadIdx.x == 0) g1x_store(block_sum + blockIdx.x, acc);
}

// window_sum[w] = sum of the window's block partials
__global__ __launch_bounds__(256) void msm_window_reduce_kernel(const G1X* __restrict__ block_sum, uint32_t per_window,
                                                                G1X* __restrict__ window_sum) {
    __shared__ G1X sh[256];
    G1X acc = G1X::identity();
    for (uint32_t i = threadIdx.x; i < per_window; i += 256) {
        const G1X p = g1x_load(block_sum + blockIdx.x * per_window + i);
        g1x_add(acc, p);
    }
    block_tree_sum(sh, acc, 256);
    if (threadIdx.x == 0) g1x_store(window_sum + blockIdx.x, acc);
}

// ------------------------------------------------------------------ host ---

size_t msm_workspace_bytes(size_t max_n, uint32_t c) {
    const uint32_t nwin = nwin_for(c);
    const size_t nbt = (size_t)nwin << (c - 1);
    const size_t ent = max_n * nwin;
    const size_t slots = 2 * ((ent + 31) / 32) + 64;
    return ent * 8 + slots * (4 + sizeof(G1X)) * 2 + nbt * (8 + sizeof(G1X)) + (nbt / 256 + nwin) * sizeof(G1X);
}

#define MSM_TRY(x)                     \
    do {                               \
        hipError_t _e = (x);           \
        if (_e != hipSuccess) {        \
            if (err) *err = _e;        \
            msm_workspace_destroy(ws); \
            return nullptr;            \
        }                              \
    } while (0)

MsmWorkspace* msm_workspace_create(size_t max_n, uint32_t c, hipError_t* err) {
    if (err) *err = hipSuccess;
    if (c == 0) c = msm_auto_window(max_n);
    if (c < 9 || c > 16 || max_n == 0 || max_n > ((size_t)1 << 26)) {
        if (err) *err = hipErrorInvalidValue;
        return nullptr;
    }
    MsmWorkspace* ws = new MsmWorkspace();
    memset(ws, 0, sizeof(*ws));
    ws->max_n = max_n;
    ws->c = c;
    ws->nwin = nwin_for(c);
    ws->nb = 1u << (c - 1);
    ws->seg0 = 32;
    ws->seg1 = 16;
    const size_t nbt = (size_t)ws->nwin * ws->nb;
    const size_t ent = max_n * ws->nwin;
    ws->slot_cap = 2 * ((ent + ws->seg0 - 1) / ws->seg0) + 64;
    MSM_TRY(hipMalloc(&ws->hist, (nbt + 1) * 4));
    MSM_TRY(hipMalloc(&ws->cursor, (nbt + 1) * 4));
    MSM_TRY(hipMalloc(&ws->counts, 16 * 4));
    MSM_TRY(hipMalloc(&ws->entries, ent * sizeof(uint2)));
    for (int i = 0; i < 2; i++) {
        const size_t cap = i == 0 ? ws->slot_cap : (2 * ((ws->slot_cap + ws->seg1 - 1) / ws->seg1) + 64);
        MSM_TRY(hipMalloc(&ws->slot_bucket[i], cap * 4));
        MSM_TRY(hipMalloc(&ws->slot_pt[i], cap * sizeof(G1X)));
    }
    MSM_TRY(hipMalloc(&ws->bucket_sum, nbt * sizeof(G1X)));
    MSM_TRY(hipMalloc(&ws->block_sum, (nbt / 256) * sizeof(G1X)));
    MSM_TRY(hipMalloc(&ws->window_sum, ws->nwin * sizeof(G1X)));
    return ws;
}

void msm_workspace_destroy(MsmWorkspace* ws) {
    if (!ws) return;
    hipFree(ws->hist);
    hipFree(ws->cursor);
    hipFree(ws->counts);
    hipFree(ws->entries);
    for (int i = 0; i < 2; i++) {
        hipFree(ws->slot_bucket[i]);
        hipFree(ws->slot_pt[i]);
    }
    hipFree(ws->bucket_sum);
    hipFree(ws->block_sum);
    hipFree(ws->window_sum);
    delete ws;
}

hipError_t msm_run(MsmWorkspace* ws, const Fr* scalars, const G1Affine* bases, size_t n, hipStream_t st,
                   G1X* host_window_sums, uint32_t* nwin_out, uint32_t* c_out, hipEvent_t* accum_events) {
    if (n > ws->max_n) return hipErrorInvalidValue;
    const uint32_t c = ws->c, nwin = ws->nwin, nb = ws->nb;
    const uint32_t nbt = nwin * nb;
    *nwin_out = nwin;
    *c_out = c;
    hipError_t e;
    if ((e = hipMemsetAsync(ws->hist, 0, (nbt + 1) * 4, st)) != hipSuccess) return e;
    hipLaunchKernelGGL(msm_clear_buckets_kernel, dim3((nbt + 255) / 256), dim3(256), 0, st, ws->bucket_sum, nbt);
    if (n > 0) {
        const uint32_t nblk = (uint32_t)((n + 255) / 256);
        hipLaunchKernelGGL(msm_digits_kernel<false>, dim3(nblk), dim3(256), 0, st, scalars, (uint32_t)n, c, nwin,
                           ws->hist, (uint2*)nullptr);
        hipLaunchKernelGGL(msm_scan_kernel, dim3(1), dim3(1024), 0, st, ws->hist, ws->cursor, nbt, ws->counts);
        hipLaunchKernelGGL(msm_digits_kernel<true>, dim3(nblk), dim3(256), 0, st, scalars, (uint32_t)n, c, nwin,
                           ws->cursor, ws->entries);
        // level 0
        size_t worst = (size_t)n * nwin;  // worst-case entry count
        size_t threads = (worst + ws->seg0 - 1) / ws->seg0;
        if (accum_events) hipEventRecord(accum_events[0], st);
        hipLaunchKernelGGL(msm_accumulate_kernel, dim3((uint32_t)((threads + 63) / 64)), dim3(64), 0, st, ws->entries,
                           bases, ws->counts, ws->seg0, ws->bucket_sum, ws->slot_bucket[0], ws->slot_pt[0],
                           ws->counts + 1);
        if (accum_events) hipEventRecord(accum_events[1], st);
        size_t count = 2 * threads;  // worst-case slot count
        int cur = 0, level = 1;
        while (count > 256 && level < 14) {
            threads = (count + ws->seg1 - 1) / ws->seg1;
            hipLaunchKernelGGL(msm_reduce_slots_kernel, dim3((uint32_t)((threads + 63) / 64)), dim3(64), 0, st,
                               ws->slot_bucket[cur], ws->slot_pt[cur], ws->counts + level, ws->seg1, ws->bucket_sum,
                               ws->slot_bucket[cur ^ 1], ws->slot_pt[cur ^ 1], ws->counts + level + 1);
            count = 2 * threads;
            cur ^= 1;
            level++;
        }
        hipLaunchKernelGGL(msm_finalize_slots_kernel, dim3((uint32_t)((count + 63) / 64)), dim3(64), 0, st,
                           ws->slot_bucket[cur], ws->slot_pt[cur], ws->counts + level, ws->bucket_sum);
    }
    hipLaunchKernelGGL(msm_bucket_reduce_kernel, dim3(nbt / 256), dim3(256), 0, st, ws->bucket_sum, nb, ws->block_sum);
    hipLaunchKernelGGL(msm_window_reduce_kernel, dim3(nwin), dim3(256), 0, st, ws->block_sum, nb / 256,
                       ws->window_sum);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    return hipMemcpyAsync(host_window_sums, ws->window_sum, nwin * sizeof(G1X), hipMemcpyDeviceToHost, st);
}

G1Jac msm_finish_host(const G1X* window_sums, uint32_t nwin, uint32_t c) {
    G1X acc = G1X::identity();
    for (int w = (int)nwin - 1; w >= 0; w--) {
        for (uint32_t i = 0; i < c; i++) acc = g1x_dbl(acc);
        g1x_add(acc, window_sums[w]);
    }
    return g1x_to_jac(acc);
}

}  // namespace zk
