// prover_kernels.hip — the streaming / scan kernels between the MSMs and NTTs of
// create_proof: Montgomery conversion, ChaCha20 field sampling, lookup
// permutation (range table), permutation / lookup grand products (batch
// inversion + prefix product), linear combinations and Kate division for the
// multi-open argument.
//
// Device replacements for the host loops of halo2_proofs
// `plonk/lookup/prover.rs` (permute_expression_pair, commit_product),
// `plonk/permutation/prover.rs` (commit), `plonk/vanishing/prover.rs` (random
// poly), `arithmetic::kate_division` and the polynomial sums in
// `poly/kzg/multiopen/{gwc,shplonk}/prover.rs` (SURVEY.md §8a a9, §8f-2, §8f-3;
// reached from halo2-circuits/src/ecc/ecdsa_p256.rs:366-373,416-423).
#include <string.h>

#include "prover.h"
#include "field29.hip.h"
#include "hostutil.h"

namespace zk {

// ------------------------------------------------------------ elementwise ---

__global__ void to_mont_kernel(Fr* a, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) fe_store(a + i, fe_to_mont(fe_load(a + i)));
}
void launch_to_mont(Fr* a, uint32_t n, hipStream_t st) {
    hipLaunchKernelGGL(to_mont_kernel, dim3((n + 255) / 256), dim3(256), 0, st, a, n);
}

__global__ void mul_kernel(Fr* out, const Fr* a, const Fr* b, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) fe_store(out + i, fe_mul(fe_load(a + i), fe_load(b + i)));
}
void launch_mul(Fr* out, const Fr* a, const Fr* b, uint32_t n, hipStream_t st) {
    hipLaunchKernelGGL(mul_kernel, dim3((n + 255) / 256), dim3(256), 0, st, out, a, b, n);
}

// out[i] = sum_j c[j] * in[j][i]  (+ out[i] if accumulate); in[j] may be shorter than n (zero-padded).
// On the carry-free field (field29.hip.h), lazily: a value is taken as the plain limbs of its standard form (which is the
// internal form of v / 32), the coefficients arrive times 32, so that a product is the standard form of c v again — nothing
// to convert — and four products share one Montgomery reduction.  The sum stays below 168 p (at most 40 inputs) and one
// last product by 2^261 (the identity of this product) brings it below 2p.
__global__ __launch_bounds__(256) void lincomb_kernel(LincombArgs a) {
    typedef Fe29<FrParams> Fr29;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    Fr29 zero;
#pragma unroll
    for (int l = 0; l < 9; l++) zero.l[l] = 0;
    auto get = [&](uint32_t j) { return i < a.len[j] ? to29(fe_load(a.in[j] + i)) : zero; };  // (1 ; 29)
    Fr29 acc = a.accumulate ? to29(fe_load(a.out + i)) : zero;
    uint32_t j = 0;
    // units first come as they are; products in groups of four (the host orders nothing: runs are taken as they come)
#pragma unroll 1
    while (j < a.count) {
        if (a.unit[j]) {
            acc = norm29(add29(acc, get(j)));
            j++;
        } else if (j + 4 <= a.count && !a.unit[j + 1] && !a.unit[j + 2] && !a.unit[j + 3]) {
            const Fr29 v[4] = {get(j), get(j + 1), get(j + 2), get(j + 3)};
            const Fr29 c[4] = {to29(a.c[j]), to29(a.c[j + 1]), to29(a.c[j + 2]), to29(a.c[j + 3])};
            acc = norm29(add29(acc, mulKadd29<4>(v, c)));   // 4 <= 168: < 2p
            j += 4;
        } else {
            acc = norm29(add29(acc, mul29(get(j), to29(a.c[j]))));
            j++;
        }
    }
    // at most 40 terms of about p each + the old value + 2p per subtraction: below 48 p
    if (i == 0 && a.sub0) acc = norm29(sub29<2, 29>(acc, to29(a.sub0_val)));
    if (i < a.sub_low_n) acc = norm29(sub29<2, 29>(acc, to29(a.sub_low[i])));
    Fr r = from29(mul29(acc, const_pow2_29<261, FrParams>()));  // value bound <= 48
    reduce_once(r);
    fe_store(a.out + i, r);
}
void launch_lincomb(const LincombArgs& a, hipStream_t st) {
    LincombArgs b = a;
    const Fr k32 = fr_from_u64(32);
    for (uint32_t j = 0; j < b.count; j++)
        if (!b.unit[j]) b.c[j] = fe_mul(b.c[j], k32);  // the kernel's products divide by 32 (see above)
    hipLaunchKernelGGL(lincomb_kernel, dim3((b.n + 255) / 256), dim3(256), 0, st, b);
}

// The same sum for hundreds of inputs (the many-column shapes open 350 polynomials of a few thousand coefficients each):
// the argument list is read from device memory, 2^log_ns lanes share a coefficient — lane s takes terms s, s + ns, ... in
// groups of four per reduction — and the shares are added through LDS.  One launch instead of one per 40 inputs.
__global__ __launch_bounds__(256) void lincomb_terms_kernel(const LcTerm* __restrict__ terms, uint32_t count, Fr* __restrict__ out,
                                                            uint32_t log_ns, uint32_t sub0, Fr sub0_val, uint32_t low_n, LcLow low) {
    typedef Fe29<FrParams> Fr29;
    __shared__ uint32_t part[256 * 9];
    const uint32_t ns = 1u << log_ns, rows = 256u >> log_ns;
    const uint32_t row = threadIdx.x & (rows - 1), sl = threadIdx.x >> (8 - log_ns);
    const uint32_t i = blockIdx.x * rows + row;  // n is a multiple of 256
    const Fr29 id = const_pow2_29<261, FrParams>();  // the identity of this product (see lincomb_kernel)
    Fr29 acc;
#pragma unroll
    for (int l = 0; l < 9; l++) acc.l[l] = 0;
    uint32_t j = sl, groups = 0;
#pragma unroll 1
    for (; j + 3 * ns < count; j += 4 * ns) {
        const Fr29 v[4] = {to29(fe_load(terms[j].in + i)), to29(fe_load(terms[j + ns].in + i)),
                           to29(fe_load(terms[j + 2 * ns].in + i)), to29(fe_load(terms[j + 3 * ns].in + i))};
        const Fr29 c[4] = {to29(fe_load(&terms[j].c)), to29(fe_load(&terms[j + ns].c)), to29(fe_load(&terms[j + 2 * ns].c)),
                           to29(fe_load(&terms[j + 3 * ns].c))};
        acc = norm29(add29(acc, mulKadd29<4>(v, c)));  // + 2p
        if (++groups == 64) {                          // uniform: the sum goes back below 2p
            acc = mul29(acc, id);
            groups = 0;
        }
    }
#pragma unroll 1
    for (; j < count; j += ns) acc = norm29(add29(acc, mul29(to29(fe_load(terms[j].in + i)), to29(fe_load(&terms[j].c)))));
    const Fr29 share = mul29(acc, id);  // <= 136 p -> (2 ; 29)
#pragma unroll
    for (int l = 0; l < 9; l++) part[l * 256 + threadIdx.x] = share.l[l];
    __syncthreads();
    if (sl) return;
    Fr29 total = share;
    for (uint32_t s = 1; s < ns; s++) {
        Fr29 v;
#pragma unroll
        for (int l = 0; l < 9; l++) v.l[l] = part[l * 256 + s * rows + row];
        total = norm29(add29(total, v));  // <= 32 p
    }
    if (i == 0 && sub0) total = norm29(sub29<2, 29>(total, to29(sub0_val)));
    if (i < low_n) total = norm29(sub29<2, 29>(total, to29(low.v[i])));
    Fr r = from29(mul29(total, id));
    reduce_once(r);
    fe_store(out + i, r);
}
void launch_lincomb_terms(LcTerm* h_terms, LcTerm* d_terms, uint32_t count, Fr* out, uint32_t n, bool sub0, const Fr& sub0_val,
                          const Fr* low, uint32_t low_n, hipStream_t st) {
    const Fr k32 = fr_from_u64(32);
    for (uint32_t j = 0; j < count; j++) h_terms[j].c = fe_mul(h_terms[j].c, k32);  // the kernel's products divide by 32
    hipMemcpyAsync(d_terms, h_terms, (size_t)count * sizeof(LcTerm), hipMemcpyHostToDevice, st);
    uint32_t lg = 0;
    while ((1u << (lg + 1)) <= n) lg++;
    uint32_t log_ns = 0;  // lanes per coefficient: ~2^16 lanes in all, at most 16, at least 8 terms each
    while (log_ns < 4 && lg + log_ns < 16 && (count >> (log_ns + 1)) >= 8) log_ns++;
    LcLow lw;
    memset(&lw, 0, sizeof(lw));
    for (uint32_t t = 0; t < low_n && t < 8; t++) lw.v[t] = low[t];
    hipLaunchKernelGGL(lincomb_terms_kernel, dim3(n >> (8 - log_ns)), dim3(256), 0, st, d_terms, count, out, log_ns, sub0 ? 1u : 0u,
                       sub0_val, low_n, lw);
}

__global__ void scale_kernel(Fr* a, Fr c, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) fe_store(a + i, fe_mul(fe_load(a + i), c));
}
void launch_scale(Fr* a, const Fr& c, uint32_t n, hipStream_t st) {
    hipLaunchKernelGGL(scale_kernel, dim3((n + 255) / 256), dim3(256), 0, st, a, c, n);
}

// -------------------------------------------------------------- ChaCha20 ----
// out[i] = Fr::from_u512(keystream block (start + i)) — one Fr::random per 64-byte block.
__device__ __forceinline__ uint32_t rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
#define QR(a, b, c, d)                 \
    a += b; d ^= a; d = rotl(d, 16);   \
    c += d; b ^= c; b = rotl(b, 12);   \
    a += b; d ^= a; d = rotl(d, 8);    \
    c += d; b ^= c; b = rotl(b, 7);

__global__ void chacha_fr_kernel(ChaChaKey key, uint64_t start_block, Fr* out, uint32_t count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint64_t ctr = start_block + i;
    uint32_t s[16] = {0x61707865, 0x3320646e, 0x79622d32, 0x6b206574};
#pragma unroll
    for (int k = 0; k < 8; k++) s[4 + k] = key.w[k];
    s[12] = (uint32_t)ctr;
    s[13] = (uint32_t)(ctr >> 32);
    s[14] = 0;
    s[15] = 0;
    uint32_t w[16];
#pragma unroll
    for (int k = 0; k < 16; k++) w[k] = s[k];
#pragma unroll 1
    for (int r = 0; r < 10; r++) {
        QR(w[0], w[4], w[8], w[12]) QR(w[1], w[5], w[9], w[13]) QR(w[2], w[6], w[10], w[14]) QR(w[3], w[7], w[11], w[15])
        QR(w[0], w[5], w[10], w[15]) QR(w[1], w[6], w[11], w[12]) QR(w[2], w[7], w[8], w[13]) QR(w[3], w[4], w[9], w[14])
    }
    Fr lo, hi;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        lo.v[k] = w[k] + s[k];
        hi.v[k] = w[8 + k] + s[8 + k];
    }
    const Fr r2 = Fr::r2();
    const Fr r3 = fe_mul(r2, r2);
    fe_store(out + i, fe_add(fe_mul(r2, lo), fe_mul(r3, hi)));
}
void launch_chacha_fr(const ChaChaKey& key, uint64_t start_block, Fr* out, uint32_t count, hipStream_t st) {
    hipLaunchKernelGGL(chacha_fr_kernel, dim3((count + 255) / 256), dim3(256), 0, st, key, start_block, out, count);
}

// ------------------------------------------------------------- u32 scans ----
// exclusive scan of in[0..m) -> out[0..m], out[m] = total.  One block; m up to a few million.
__global__ __launch_bounds__(1024) void scan_u32_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, uint32_t m) {
    __shared__ uint32_t part[1024];
    const uint32_t chunk = (m + 1023) / 1024;
    const uint32_t lo = min(m, threadIdx.x * chunk), hi = min(m, lo + chunk);
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; i++) sum += in[i];
    part[threadIdx.x] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        const uint32_t v = (threadIdx.x >= d) ? part[threadIdx.x - d] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - sum;
    for (uint32_t i = lo; i < hi; i++) {
        const uint32_t h = in[i];
        out[i] = run;
        run += h;
    }
    if (threadIdx.x == 1023) out[m] = part[1023];
}
void launch_scan_u32(const uint32_t* in, uint32_t* out, uint32_t m, hipStream_t st) {
    hipLaunchKernelGGL(scan_u32_kernel, dim3(1), dim3(1024), 0, st, in, out, m);
}

// ------------------------------------------------ lookup permutation --------
// Range-table specialisation of permute_expression_pair: table column = 0..T-1 in
// rows 0..T-1 and zeros up to the usable rows (halo2-lib RangeConfig; known answer
// K2).  hist[v] = multiplicity of v in the first `usable` input rows; err set if an
// input is not a table element (halo2: Error::ConstraintSystemFailure).
// All lookups of a proof go through these kernels together: blockIdx.y = lookup, whose scratch arrays sit
// `stride` words after the previous lookup's.
__global__ __launch_bounds__(256) void lk_hist_kernel(LkPtrs ptrs, uint32_t usable, uint32_t T, uint32_t* __restrict__ hist0,
                                                      uint32_t stride, uint32_t* __restrict__ err) {
    const Fr* __restrict__ inp = ptrs.inp[blockIdx.y];
    uint32_t* __restrict__ hist = hist0 + (size_t)blockIdx.y * stride;
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t val = 0xffffffffu;  // not a table element / out of range
    if (i < usable) {
        const Fr v = fe_from_mont(fe_load(inp + i));
        uint32_t hi = 0;
#pragma unroll
        for (int k = 1; k < 8; k++) hi |= v.v[k];
        if (hi || v.v[0] >= T) atomicOr(err, 1u);
        else val = v.v[0];
    }
    // unselected rows contribute q_lookup * a = 0: most of the column hits hist[0].
    // Count zeros per wave with a ballot and issue one atomic for them.
    const unsigned long long zmask = __ballot(val == 0);
    if (val == 0) {
        if ((threadIdx.x & 63) == (uint32_t)__ffsll((long long)zmask) - 1) atomicAdd(&hist[0], (uint32_t)__popcll(zmask));
    } else if (val != 0xffffffffu) {
        atomicAdd(&hist[val], 1u);
    }
}

// present[v] = hist[v] > 0 ; absent[v] = (v >= 1 && hist[v] == 0)
__global__ void lk_flags_kernel(const uint32_t* __restrict__ hist0, uint32_t T, uint32_t* __restrict__ present0,
                                uint32_t* __restrict__ absent0, uint32_t stride) {
    const uint32_t* __restrict__ hist = hist0 + (size_t)blockIdx.y * stride;
    uint32_t* __restrict__ present = present0 + (size_t)blockIdx.y * stride;
    uint32_t* __restrict__ absent = absent0 + (size_t)blockIdx.y * stride;
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= T) return;
    const uint32_t h = hist[v];
    present[v] = h > 0;
    absent[v] = (v >= 1 && h == 0);
}

__device__ __forceinline__ Fr small_to_mont(uint32_t v) {
    Fr a = Fr::zero();
    a.v[0] = v;
    return fe_to_mont(a);
}

// off: exclusive scan of hist (T+1); dex: exclusive scan of present (T+1); aex: exclusive scan of absent (T+1)
__global__ void lk_fill_kernel(uint32_t usable, uint32_t T, const uint32_t* __restrict__ hist0,
                               const uint32_t* __restrict__ off0, const uint32_t* __restrict__ dex0,
                               const uint32_t* __restrict__ aex0, uint32_t stride, LkPtrs ptrs) {
    const uint32_t* __restrict__ hist = hist0 + (size_t)blockIdx.y * stride;
    const uint32_t* __restrict__ off = off0 + (size_t)blockIdx.y * stride;
    const uint32_t* __restrict__ dex = dex0 + (size_t)blockIdx.y * stride;
    const uint32_t* __restrict__ aex = aex0 + (size_t)blockIdx.y * stride;
    Fr* __restrict__ ap = ptrs.ap[blockIdx.y];
    Fr* __restrict__ sp = ptrs.sp[blockIdx.y];
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= usable) return;
    // largest v with off[v] <= p
    uint32_t lo = 0, hi = T - 1;
    while (lo < hi) {
        const uint32_t mid = (lo + hi + 1) >> 1;
        if (off[mid] <= p) lo = mid; else hi = mid - 1;
    }
    const uint32_t v = lo;
    fe_store(ap + p, small_to_mont(v));
    if (p == off[v]) {
        fe_store(sp + p, small_to_mont(v));
        return;
    }
    // repeated row: index j among repeated rows (ascending) = p - (#present values <= v)
    const uint32_t D = dex[v] + 1;           // present values <= v (v itself is present)
    const uint32_t j = p - D;
    const uint32_t m = usable - dex[T];      // number of repeated rows = number of leftover table cells
    const uint32_t q = m - 1 - j;            // leftovers ascending are handed to repeated rows descending
    const uint32_t c0 = (usable - T + 1) - (hist[0] > 0 ? 1u : 0u);  // leftover zeros
    if (q < c0) {
        fe_store(sp + p, Fr::zero());
        return;
    }
    const uint32_t t = q - c0 + 1;           // t-th absent value (1-based) among 1..T-1
    // smallest v' with (#absent in [0..v']) >= t, i.e. aex[v'+1] >= t
    uint32_t a = 1, b = T - 1;
    while (a < b) {
        const uint32_t mid = (a + b) >> 1;
        if (aex[mid + 1] >= t) b = mid; else a = mid + 1;
    }
    fe_store(sp + p, small_to_mont(a));
}

// Three exclusive scans over T entries at once (hist, present, absent), multi-block:
// (1) per-block sums, (2) one small block scans the block sums, (3) block-local scan + offset.
static constexpr uint32_t S3_BLOCK = 1024;  // elements per block (256 threads x 4)

__global__ __launch_bounds__(256) void scan3_sums_kernel(const uint32_t* __restrict__ b0, const uint32_t* __restrict__ b1,
                                                         const uint32_t* __restrict__ b2, uint32_t m,
                                                         uint32_t* __restrict__ bsum0 /* [3][nblocks] */, uint32_t nblocks,
                                                         uint32_t stride) {
    const uint32_t* __restrict__ a0 = b0 + (size_t)blockIdx.y * stride;
    const uint32_t* __restrict__ a1 = b1 + (size_t)blockIdx.y * stride;
    const uint32_t* __restrict__ a2 = b2 + (size_t)blockIdx.y * stride;
    uint32_t* __restrict__ bsum = bsum0 + (size_t)blockIdx.y * stride;
    __shared__ uint32_t sh[3][256];
    const uint32_t base = blockIdx.x * S3_BLOCK + threadIdx.x * 4;
    uint32_t s0 = 0, s1 = 0, s2 = 0;
    for (uint32_t k = 0; k < 4; k++)
        if (base + k < m) {
            s0 += a0[base + k];
            s1 += a1[base + k];
            s2 += a2[base + k];
        }
    sh[0][threadIdx.x] = s0;
    sh[1][threadIdx.x] = s1;
    sh[2][threadIdx.x] = s2;
    __syncthreads();
    for (uint32_t d = 128; d > 0; d >>= 1) {
        if (threadIdx.x < d)
            for (int q = 0; q < 3; q++) sh[q][threadIdx.x] += sh[q][threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x < 3) bsum[threadIdx.x * nblocks + blockIdx.x] = sh[threadIdx.x][0];
}

__global__ __launch_bounds__(1024) void scan3_top_kernel(uint32_t* __restrict__ bsum0, uint32_t nblocks, uint32_t stride) {
    // exclusive scan of each of the 3 rows in place; nblocks <= 1024 * 8; blockIdx.x = lookup
    uint32_t* __restrict__ bsum = bsum0 + (size_t)blockIdx.x * stride;
    __shared__ uint32_t part[1024];
    for (int q = 0; q < 3; q++) {
        uint32_t* row = bsum + q * nblocks;
        const uint32_t chunk = (nblocks + 1023) / 1024;
        const uint32_t lo = min(nblocks, threadIdx.x * chunk), hi = min(nblocks, lo + chunk);
        uint32_t sum = 0;
        for (uint32_t i = lo; i < hi; i++) sum += row[i];
        part[threadIdx.x] = sum;
        __syncthreads();
        for (uint32_t d = 1; d < 1024; d <<= 1) {
            const uint32_t v = (threadIdx.x >= d) ? part[threadIdx.x - d] : 0;
            __syncthreads();
            part[threadIdx.x] += v;
            __syncthreads();
        }
        uint32_t run = part[threadIdx.x] - sum;
        for (uint32_t i = lo; i < hi; i++) {
            const uint32_t h = row[i];
            row[i] = run;
            run += h;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void scan3_apply_kernel(const uint32_t* __restrict__ b0, const uint32_t* __restrict__ b1,
                                                          const uint32_t* __restrict__ b2, uint32_t m,
                                                          const uint32_t* __restrict__ bsum0, uint32_t nblocks,
                                                          uint32_t* __restrict__ p0, uint32_t* __restrict__ p1,
                                                          uint32_t* __restrict__ p2, uint32_t stride) {
    const size_t shift = (size_t)blockIdx.y * stride;
    const uint32_t* __restrict__ a0 = b0 + shift;
    const uint32_t* __restrict__ a1 = b1 + shift;
    const uint32_t* __restrict__ a2 = b2 + shift;
    const uint32_t* __restrict__ bsum = bsum0 + shift;
    uint32_t* __restrict__ o0 = p0 + shift;
    uint32_t* __restrict__ o1 = p1 + shift;
    uint32_t* __restrict__ o2 = p2 + shift;
    __shared__ uint32_t sh[3][256];
    const uint32_t base = blockIdx.x * S3_BLOCK + threadIdx.x * 4;
    uint32_t v[3][4];
    uint32_t s[3] = {0, 0, 0};
    for (uint32_t k = 0; k < 4; k++) {
        const bool in = base + k < m;
        v[0][k] = in ? a0[base + k] : 0;
        v[1][k] = in ? a1[base + k] : 0;
        v[2][k] = in ? a2[base + k] : 0;
        for (int q = 0; q < 3; q++) s[q] += v[q][k];
    }
    for (int q = 0; q < 3; q++) sh[q][threadIdx.x] = s[q];
    __syncthreads();
    for (uint32_t d = 1; d < 256; d <<= 1) {
        uint32_t t[3] = {0, 0, 0};
        if (threadIdx.x >= d)
            for (int q = 0; q < 3; q++) t[q] = sh[q][threadIdx.x - d];
        __syncthreads();
        for (int q = 0; q < 3; q++) sh[q][threadIdx.x] += t[q];
        __syncthreads();
    }
    const uint32_t* const ins[3] = {a0, a1, a2};
    (void)ins;
    uint32_t* const outs[3] = {o0, o1, o2};
    for (int q = 0; q < 3; q++) {
        uint32_t run = bsum[q * nblocks + blockIdx.x] + sh[q][threadIdx.x] - s[q];
        for (uint32_t k = 0; k < 4; k++) {
            if (base + k < m) outs[q][base + k] = run;
            run += v[q][k];
        }
        // total at index m (exclusive scan convention used by lk_fill)
        if (blockIdx.x == nblocks - 1 && threadIdx.x == 255) outs[q][m] = run;
    }
}

// `count` lookups at once; s: the scratch of lookup 0, lookup l's arrays `s.stride` words further each
void launch_lookup_permute(const LkPtrs& ptrs, uint32_t count, uint32_t usable, uint32_t T, LookupScratch& s, hipStream_t st) {
    hipMemsetAsync(s.hist, 0, ((size_t)(count - 1) * s.stride + T + 1) * 4, st);  // every hist (the other arrays are rewritten)
    // s.err accumulates over the lookups of a proof: the caller clears it once and reads it once
    hipLaunchKernelGGL(lk_hist_kernel, dim3((usable + 255) / 256, count), dim3(256), 0, st, ptrs, usable, T, s.hist, s.stride, s.err);
    hipLaunchKernelGGL(lk_flags_kernel, dim3((T + 255) / 256, count), dim3(256), 0, st, s.hist, T, s.present, s.absent, s.stride);
    const uint32_t nblocks = (T + S3_BLOCK - 1) / S3_BLOCK;
    hipLaunchKernelGGL(scan3_sums_kernel, dim3(nblocks, count), dim3(256), 0, st, s.hist, s.present, s.absent, T, s.bsum, nblocks,
                       s.stride);
    hipLaunchKernelGGL(scan3_top_kernel, dim3(count), dim3(1024), 0, st, s.bsum, nblocks, s.stride);
    hipLaunchKernelGGL(scan3_apply_kernel, dim3(nblocks, count), dim3(256), 0, st, s.hist, s.present, s.absent, T, s.bsum, nblocks,
                       s.off, s.dex, s.aex, s.stride);
    hipLaunchKernelGGL(lk_fill_kernel, dim3((usable + 255) / 256, count), dim3(256), 0, st, usable, T, s.hist, s.off, s.dex, s.aex,
                       s.stride, ptrs);
}

// -------------------------------------------------------- grand products ----

// den[i] = prod_c (v_c[i] + beta*sigma_c[i] + gamma);  num[i] = prod_c (v_c[i] + delta_c * w^i * beta + gamma)
__global__ void perm_numden_kernel(PermArgs a) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const Fr wi_beta = fe_mul(fe_load(a.tw + i), a.beta);
    Fr num = Fr::one(), den = Fr::one();
    for (uint32_t c = 0; c < a.ncols; c++) {
        const Fr v = fe_load(a.values[c] + i);
        const Fr vg = fe_add(v, a.gamma);
        den = fe_mul(den, fe_add(vg, fe_mul(a.beta, fe_load(a.sigma[c] + i))));
        num = fe_mul(num, fe_add(vg, fe_mul(wi_beta, a.delta[c])));
    }
    fe_store(a.num + i, num);
    fe_store(a.den + i, den);
}
void launch_perm_numden(const PermArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(perm_numden_kernel, dim3((a.n + 255) / 256), dim3(256), 0, st, a);
}
// every chunk of a proof in one launch (blockIdx.y = chunk; argument blocks in device memory): the many-column rows of
// bench_ecdsa.config have up to 176 chunks of 2^11 .. 2^13 rows — launch-bound one by one
__global__ void perm_numden_batch_kernel(const PermArgs* __restrict__ args) {
    const PermArgs& a = args[blockIdx.y];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    const Fr beta = a.beta, gamma = a.gamma;
    const Fr wi_beta = fe_mul(fe_load(a.tw + i), beta);
    Fr num = Fr::one(), den = Fr::one();
    for (uint32_t c = 0; c < a.ncols; c++) {
        const Fr v = fe_load(a.values[c] + i);
        const Fr vg = fe_add(v, gamma);
        den = fe_mul(den, fe_add(vg, fe_mul(beta, fe_load(a.sigma[c] + i))));
        num = fe_mul(num, fe_add(vg, fe_mul(wi_beta, a.delta[c])));
    }
    fe_store(a.num + i, num);
    fe_store(a.den + i, den);
}
void launch_perm_numden_batch(const PermArgs* d_args, uint32_t count, uint32_t n, hipStream_t st) {
    hipLaunchKernelGGL(perm_numden_batch_kernel, dim3((n + 255) / 256, count), dim3(256), 0, st, d_args);
}
// the same for the lookups' numerators / denominators (argument blocks: LkNumDenArgs)
__global__ void lk_numden_batch_kernel(const LkNumDenArgs* __restrict__ args, Fr beta, Fr gamma, uint32_t n) {
    const LkNumDenArgs a = args[blockIdx.y];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe_store(a.den + i, fe_mul(fe_add(fe_load(a.ap + i), beta), fe_add(fe_load(a.sp + i), gamma)));
    fe_store(a.num + i, fe_mul(fe_add(fe_load(a.inp + i), beta), fe_add(fe_load(a.tab + i), gamma)));
}
void launch_lk_numden_batch(const LkNumDenArgs* d_args, uint32_t count, const Fr& beta, const Fr& gamma, uint32_t n, hipStream_t st) {
    hipLaunchKernelGGL(lk_numden_batch_kernel, dim3((n + 255) / 256, count), dim3(256), 0, st, d_args, beta, gamma, n);
}
// dst[q][0 .. n) = src[q][0 .. n) for `count` columns in one launch (the advice columns of a request -> the prover's copies)
__global__ void copy_columns_kernel(const CopyPair* __restrict__ pairs, uint32_t n) {
    const CopyPair p = pairs[blockIdx.y];
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) fe_store(p.dst + i, fe_load(p.src + i));
}
void launch_copy_columns(const CopyPair* d_pairs, uint32_t count, uint32_t n, hipStream_t st) {
    hipLaunchKernelGGL(copy_columns_kernel, dim3((n + 255) / 256, count), dim3(256), 0, st, d_pairs, n);
}

// den = (a' + beta)(s' + gamma); num = (in + beta)(tab + gamma)
__global__ void lk_numden_kernel(const Fr* ap, const Fr* sp, const Fr* inp, const Fr* tab, Fr beta, Fr gamma, Fr* num,
                                 Fr* den, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fe_store(den + i, fe_mul(fe_add(fe_load(ap + i), beta), fe_add(fe_load(sp + i), gamma)));
    fe_store(num + i, fe_mul(fe_add(fe_load(inp + i), beta), fe_add(fe_load(tab + i), gamma)));
}
void launch_lk_numden(const Fr* ap, const Fr* sp, const Fr* inp, const Fr* tab, const Fr& beta, const Fr& gamma, Fr* num,
                      Fr* den, uint32_t n, hipStream_t st) {
    hipLaunchKernelGGL(lk_numden_kernel, dim3((n + 255) / 256), dim3(256), 0, st, ap, sp, inp, tab, beta, gamma, num, den, n);
}

// frac[i] = num[i] / den[i]  (Montgomery's trick over 16 elements per thread; 0 -> 0 like batch_invert)
__global__ void frac_kernel(const Fr* __restrict__ num, const Fr* __restrict__ den, Fr* __restrict__ frac, uint32_t n) {
    constexpr int E = 16;
    const uint32_t base = (blockIdx.x * blockDim.x + threadIdx.x) * E;
    if (base >= n) return;
    Fr d[E], pre[E];
    Fr acc = Fr::one();
#pragma unroll
    for (int k = 0; k < E; k++) {
        d[k] = (base + k < n) ? fe_load(den + base + k) : Fr::one();
        pre[k] = acc;
        if (!d[k].is_zero()) acc = fe_mul(acc, d[k]);
    }
    Fr inv = fe_inv(acc);
#pragma unroll
    for (int k = E - 1; k >= 0; k--) {
        if (base + k < n) {
            Fr r = Fr::zero();
            if (!d[k].is_zero()) {
                r = fe_mul(inv, pre[k]);
                inv = fe_mul(inv, d[k]);
            }
            fe_store(frac + base + k, fe_mul(r, fe_load(num + base + k)));
        }
    }
}
void launch_frac(const Fr* num, const Fr* den, Fr* frac, uint32_t n, hipStream_t st) {
    const uint32_t threads = (n + 15) / 16;
    hipLaunchKernelGGL(frac_kernel, dim3((threads + 63) / 64), dim3(64), 0, st, num, den, frac, n);
}

// Prefix product: z[0] = init, z[i+1] = z[i] * f[i] for i < n-1.
// pass 1: block-local inclusive products (2048 elements per block) + block totals
// pass 2: one block turns the totals into exclusive block offsets (times init)
// pass 3: z[i+1] = offset[block] * local[i]
static constexpr uint32_t PP_E = 8, PP_T = 256, PP_B = PP_E * PP_T;

__global__ __launch_bounds__(PP_T) void pp_local_kernel(const Fr* __restrict__ f, Fr* __restrict__ local, Fr* __restrict__ tot, uint32_t n) {
    __shared__ Fr sh[PP_T];
    const uint32_t base = blockIdx.x * PP_B + threadIdx.x * PP_E;
    Fr v[PP_E];
    Fr acc = Fr::one();
#pragma unroll
    for (uint32_t k = 0; k < PP_E; k++) {
        if (base + k < n) acc = fe_mul(acc, fe_load(f + base + k));
        v[k] = acc;
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t d = 1; d < PP_T; d <<= 1) {
        Fr o = Fr::one();
        const bool has = threadIdx.x >= d;
        if (has) o = sh[threadIdx.x - d];
        __syncthreads();
        if (has) sh[threadIdx.x] = fe_mul(sh[threadIdx.x], o);
        __syncthreads();
    }
    const Fr excl = threadIdx.x ? sh[threadIdx.x - 1] : Fr::one();
#pragma unroll
    for (uint32_t k = 0; k < PP_E; k++)
        if (base + k < n) fe_store(local + base + k, fe_mul(v[k], excl));
    if (threadIdx.x == PP_T - 1) fe_store(tot + blockIdx.x, sh[PP_T - 1]);
}

__global__ __launch_bounds__(1024) void pp_offsets_kernel(Fr* __restrict__ tot, uint32_t nblocks, const Fr* __restrict__ init_ptr, Fr init_val) {
    // sequential per thread chunk + block scan; nblocks <= 1024 * 64
    __shared__ Fr sh[1024];
    const Fr init = init_ptr ? fe_load(init_ptr) : init_val;
    const uint32_t chunk = (nblocks + 1023) / 1024;
    const uint32_t lo = min(nblocks, threadIdx.x * chunk), hi = min(nblocks, lo + chunk);
    Fr acc = Fr::one();
    for (uint32_t i = lo; i < hi; i++) acc = fe_mul(acc, fe_load(tot + i));
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        Fr o = Fr::one();
        const bool has = threadIdx.x >= d;
        if (has) o = sh[threadIdx.x - d];
        __syncthreads();
        if (has) sh[threadIdx.x] = fe_mul(sh[threadIdx.x], o);
        __syncthreads();
    }
    Fr run = fe_mul(init, threadIdx.x ? sh[threadIdx.x - 1] : Fr::one());
    for (uint32_t i = lo; i < hi; i++) {
        const Fr t = fe_load(tot + i);
        fe_store(tot + i, run);  // exclusive offset (includes init)
        run = fe_mul(run, t);
    }
}

__global__ __launch_bounds__(PP_T) void pp_apply_kernel(const Fr* __restrict__ local, const Fr* __restrict__ offs, Fr* __restrict__ z, uint32_t n) {
    const uint32_t i = blockIdx.x * PP_T + threadIdx.x;
    if (i >= n) return;
    const Fr off = fe_load(offs + i / PP_B);
    if (i == 0) fe_store(z, off);  // = init
    if (i + 1 < n) fe_store(z + i + 1, fe_mul(off, fe_load(local + i)));
}

void launch_prefix_product(const Fr* f, Fr* z, uint32_t n, const Fr* init_dev, const Fr& init_val, Fr* tmp_local, Fr* tmp_tot, hipStream_t st) {
    const uint32_t nblocks = (n + PP_B - 1) / PP_B;
    hipLaunchKernelGGL(pp_local_kernel, dim3(nblocks), dim3(PP_T), 0, st, f, tmp_local, tmp_tot, n);
    hipLaunchKernelGGL(pp_offsets_kernel, dim3(1), dim3(1024), 0, st, tmp_tot, nblocks, init_dev, init_val);
    hipLaunchKernelGGL(pp_apply_kernel, dim3((n + PP_T - 1) / PP_T), dim3(PP_T), 0, st, tmp_local, tmp_tot, z, n);
}

// ---- grand products without per-element inversions ------------------------------------------------
// z[0] = init, z[i+1] = z[i] * num[i] / den[i]  ==>  z[i+1] = init * P_i * R_{i+1} / Q, with
// P_i = prod_{j<=i} num_j, R_i = prod_{j>=i} den_j, Q = R_0.  Two block scans (one forward, one
// backward) and ONE field inversion (on the host) replace n batched inversions, whose Fermat chains
// left the chip latency-bound.  A zero denominator (probability ~2^-230 under random beta, gamma)
// makes Q zero: the caller then takes the batch-inversion path (launch_frac + launch_prefix_product),
// which maps 0 -> 0 as halo2's batch_invert does.
// ---- all grand products of a proof at once -------------------------------------------------------
// The same scheme with blockIdx.y = product: ONE host round trip (all Q's out, all inverses back) for
// every permutation chunk and lookup of the proof.  A chained product (permutation chunk ci > 0) starts
// from its predecessor's value at row `usable`; that value is a product of scan outputs, so a tiny kernel
// walks the chain before the apply pass.
__global__ __launch_bounds__(PP_T) void gp_local_batch_kernel(const GpItem* __restrict__ items, uint32_t n) {
    const GpItem it = items[blockIdx.y];
    __shared__ Fr shp[PP_T], shr[PP_T];
    const uint32_t base = blockIdx.x * PP_B + threadIdx.x * PP_E;
    Fr v[PP_E], w[PP_E];
    Fr acc = Fr::one();
#pragma unroll
    for (uint32_t k = 0; k < PP_E; k++) {
        if (base + k < n) acc = fe_mul(acc, fe_load(it.num + base + k));
        v[k] = acc;
    }
    shp[threadIdx.x] = acc;
    acc = Fr::one();
#pragma unroll
    for (int k = PP_E - 1; k >= 0; k--) {
        if (base + k < n) acc = fe_mul(acc, fe_load(it.den + base + k));
        w[k] = acc;
    }
    shr[threadIdx.x] = acc;
    __syncthreads();
#pragma unroll 1
    for (uint32_t d = 1; d < PP_T; d <<= 1) {
        Fr op = Fr::one(), orr = Fr::one();
        const bool hp = threadIdx.x >= d, hr = threadIdx.x + d < PP_T;
        if (hp) op = shp[threadIdx.x - d];
        if (hr) orr = shr[threadIdx.x + d];
        __syncthreads();
        if (hp) shp[threadIdx.x] = fe_mul(shp[threadIdx.x], op);
        if (hr) shr[threadIdx.x] = fe_mul(shr[threadIdx.x], orr);
        __syncthreads();
    }
    const Fr ep = threadIdx.x ? shp[threadIdx.x - 1] : Fr::one();
    const Fr er = threadIdx.x + 1 < PP_T ? shr[threadIdx.x + 1] : Fr::one();
#pragma unroll
    for (uint32_t k = 0; k < PP_E; k++)
        if (base + k < n) {
            fe_store(it.loc_p + base + k, fe_mul(v[k], ep));
            fe_store(it.loc_r + base + k, fe_mul(w[k], er));
        }
    if (threadIdx.x == PP_T - 1) fe_store(it.tot_p + blockIdx.x, shp[PP_T - 1]);
    if (threadIdx.x == 0) fe_store(it.tot_r + blockIdx.x, shr[0]);
}

// one workgroup per product: block totals -> exclusive offsets (forward for P, backward for R); q_out[product] = Q
__global__ __launch_bounds__(256) void gp_offsets_batch_kernel(const GpItem* __restrict__ items, uint32_t nblocks,
                                                               Fr* __restrict__ q_out) {
    const GpItem it = items[blockIdx.x];
    __shared__ Fr shp[256], shr[256];
    const uint32_t chunk = (nblocks + 255) / 256;
    const uint32_t lo = min(nblocks, threadIdx.x * chunk), hi = min(nblocks, lo + chunk);
    Fr ap = Fr::one(), ar = Fr::one();
    for (uint32_t i = lo; i < hi; i++) {
        ap = fe_mul(ap, fe_load(it.tot_p + i));
        ar = fe_mul(ar, fe_load(it.tot_r + i));
    }
    shp[threadIdx.x] = ap;
    shr[threadIdx.x] = ar;
    __syncthreads();
#pragma unroll 1
    for (uint32_t d = 1; d < 256; d <<= 1) {
        Fr op = Fr::one(), orr = Fr::one();
        const bool hp = threadIdx.x >= d, hr = threadIdx.x + d < 256;
        if (hp) op = shp[threadIdx.x - d];
        if (hr) orr = shr[threadIdx.x + d];
        __syncthreads();
        if (hp) shp[threadIdx.x] = fe_mul(shp[threadIdx.x], op);
        if (hr) shr[threadIdx.x] = fe_mul(shr[threadIdx.x], orr);
        __syncthreads();
    }
    if (threadIdx.x == 0) fe_store(q_out + blockIdx.x, shr[0]);
    Fr run = threadIdx.x ? shp[threadIdx.x - 1] : Fr::one();
    for (uint32_t i = lo; i < hi; i++) {
        const Fr t = fe_load(it.tot_p + i);
        fe_store(it.tot_p + i, run);
        run = fe_mul(run, t);
    }
    run = threadIdx.x + 1 < 256 ? shr[threadIdx.x + 1] : Fr::one();
    for (uint32_t i = hi; i-- > lo;) {
        const Fr t = fe_load(it.tot_r + i);
        fe_store(it.tot_r + i, run);
        run = fe_mul(run, t);
    }
}

// k[p] = init_p / Q_p with init_p = 1, or (chained) the predecessor's z at row `row` (1 <= row < n):
// z_p[row] = k[p] * P_{row-1} * R_row.  A chain is a run of products whose members after the first carry the chain flag (the
// permutation chunks of one proof); a list may hold several chains (the proofs of a lock-step batch):
// init_p = prod over the chain's members j < p of (P_{j,row-1} R_{j,row} / Q_j) — one lane per product and a SEGMENTED block
// scan (a product without the flag starts a new segment).  nprod <= 256.
__global__ __launch_bounds__(256) void gp_chain_kernel(const GpItem* __restrict__ items, const Fr* __restrict__ q_inv, uint32_t nprod,
                                                       uint32_t row, Fr* __restrict__ kout, Fr* __restrict__ init_out) {
    __shared__ Fr sh[256];
    __shared__ uint32_t head[256];  // the segment of lane p has a head at or before p within the scanned distance
    const uint32_t p = threadIdx.x;
    Fr v = Fr::one(), qi = Fr::one();
    bool chained_next = false;  // does product p + 1 start from this one?
    bool is_head = true;
    if (p < nprod) {
        const GpItem it = items[p];
        qi = fe_load(q_inv + p);
        is_head = p == 0 || !it.chain;
        chained_next = p + 1 < nprod && items[p + 1].chain;
        if (chained_next) {
            const Fr pfx = fe_mul(fe_load(it.tot_p + (row - 1) / PP_B), fe_load(it.loc_p + row - 1));
            const Fr sfx = fe_mul(fe_load(it.tot_r + row / PP_B), fe_load(it.loc_r + row));
            v = fe_mul(qi, fe_mul(pfx, sfx));
        }
    }
    sh[p] = v;  // 1 outside a chain: the inclusive scan below is then the chain's running product
    head[p] = is_head ? 1u : 0u;
    __syncthreads();
#pragma unroll 1
    for (uint32_t d = 1; d < 256; d <<= 1) {
        Fr o = Fr::one();
        uint32_t oh = 0;
        const bool has = p >= d && !head[p];  // a lane whose segment head has been reached takes nothing from beyond it
        if (p >= d) oh = head[p - d];
        if (has) o = sh[p - d];
        __syncthreads();
        if (has) {
            sh[p] = fe_mul(sh[p], o);
            head[p] = oh;
        }
        __syncthreads();
    }
    if (p < nprod) {
        const Fr init = (p && items[p].chain) ? sh[p - 1] : Fr::one();
        fe_store(init_out + p, init);
        fe_store(kout + p, fe_mul(init, qi));
    }
}

__global__ __launch_bounds__(PP_T) void gp_apply_batch_kernel(const GpItem* __restrict__ items, const Fr* __restrict__ kk,
                                                              const Fr* __restrict__ init, uint32_t n) {
    const GpItem it = items[blockIdx.y];
    const uint32_t i = blockIdx.x * PP_T + threadIdx.x;
    if (i >= n) return;
    if (i == 0) fe_store(it.z, fe_load(init + blockIdx.y));
    if (i + 1 >= n) return;
    const Fr pfx = fe_mul(fe_mul(fe_load(it.tot_p + i / PP_B), fe_load(kk + blockIdx.y)), fe_load(it.loc_p + i));
    const Fr sfx = fe_mul(fe_load(it.tot_r + (i + 1) / PP_B), fe_load(it.loc_r + i + 1));
    fe_store(it.z + i + 1, fe_mul(pfx, sfx));
}

uint32_t gp_blocks(uint32_t n) { return (n + PP_B - 1) / PP_B; }

// Phase 1: scans of every product; q_dev[p] = product of product p's denominators.
void launch_gp_batch_scan(const GpItem* d_items, uint32_t nprod, uint32_t n, Fr* q_dev, hipStream_t st) {
    const uint32_t nblocks = gp_blocks(n);
    hipLaunchKernelGGL(gp_local_batch_kernel, dim3(nblocks, nprod), dim3(PP_T), 0, st, d_items, n);
    hipLaunchKernelGGL(gp_offsets_batch_kernel, dim3(nprod), dim3(256), 0, st, d_items, nblocks, q_dev);
}
// Phase 2 (after the host inverted the q's into q_inv_dev): chain constants, then every z.
void launch_gp_batch_apply(const GpItem* d_items, uint32_t nprod, uint32_t n, uint32_t chain_row, const Fr* q_inv_dev, Fr* k_dev,
                           Fr* init_dev, hipStream_t st) {
    hipLaunchKernelGGL(gp_chain_kernel, dim3(1), dim3(256), 0, st, d_items, q_inv_dev, nprod, chain_row, k_dev, init_dev);
    hipLaunchKernelGGL(gp_apply_batch_kernel, dim3((n + PP_T - 1) / PP_T, nprod), dim3(PP_T), 0, st, d_items, k_dev, init_dev, n);
}

// ---- staged row writes: blinding rows of many columns in one upload + one launch --------------------
__global__ __launch_bounds__(64) void scatter_rows_kernel(const RowEntry* __restrict__ e) {
    const RowEntry* r = e + blockIdx.x;
    if (threadIdx.x < r->count) fe_store(r->dst + threadIdx.x, fe_load(r->vals + threadIdx.x));
}
void launch_scatter_rows(const RowEntry* d_entries, uint32_t count, hipStream_t st) {
    hipLaunchKernelGGL(scatter_rows_kernel, dim3(count), dim3(64), 0, st, d_entries);
}

// ---------------------------------------------------------- Kate division ---
// q = (p - p(z)) / (X - z):  q[i-1] = p[i] + z*q[i], q[n-1] = 0.  Chunks of KD_L
// coefficients per thread: (1) chunk value c_t = sum_i p[tL+i] z^i, (2) suffix Horner
// over chunks carry_t = c_{t+1} + z^L carry_{t+1}, (3) replay each chunk from its carry.
// Up to KD_MAX_BATCH independent divisions per launch (blockIdx.y): SHPLONK divides every rotation
// set's polynomial by that set's next point in the same step.  q may be p (in place).
#ifndef ZK_KD_L
#define ZK_KD_L 8
#endif
// shortest chunk (power of two): 8: 4x the lanes of 32 (a 2^19 division is 65 536 Horner chains of 8 instead of 16 384 of 32:
// -0.1 ms per k=19 proof, -0.3 ms at k=15..17)
static constexpr uint32_t KD_L_MIN = ZK_KD_L;
static constexpr uint32_t KD_BLK = 256;
static constexpr uint32_t KD_TOP = 1024;  // most blocks of a division: every block sums the aggregates above it directly
// chunk length of an n-coefficient division: the shortest that keeps the number of blocks within KD_TOP (8 up to 2^21,
// 16 at 2^22, ...)
__host__ __device__ inline uint32_t kd_len(uint32_t n) {
    uint32_t L = KD_L_MIN;
    while (((n + L - 1) / L + KD_BLK - 1) / KD_BLK > KD_TOP) L <<= 1;
    return L;
}

struct KdPtrs {
    const Fr* p[KD_MAX_BATCH];
    Fr* q[KD_MAX_BATCH];
    Fr z[KD_MAX_BATCH];
};
struct KdStep {
    Fr step[KD_MAX_BATCH][8];  // Z^(2^j), j = 0..7, Z = z^KD_L
};
struct KdTop {
    Fr top[KD_MAX_BATCH][10];  // (Z^256)^(2^j): exponents below KD_TOP = 2^10
};

// per-division scratch: (m unused) | suf[m] | agg[nblk] | tpow[nblk] | zpow[256]
__host__ __device__ inline uint32_t kd_scratch_elems(uint32_t n) {
    const uint32_t L = kd_len(n), m = (n + L - 1) / L, nblk = (m + KD_BLK - 1) / KD_BLK;
    return 2 * m + 2 * nblk + 256;
}
uint32_t kate_division_scratch(uint32_t n) { return kd_scratch_elems(n); }

// Carries between chunks: K_t = sum_{s > t} c_s Z^(s-t-1), Z = z^KD_L.  Two-level suffix scan with KNOWN multipliers (powers
// of Z), so a Hillis-Steele step is one product.  Two launches per batch of divisions:
//   kd_scan   workgroup B < nblk: the chunk values c_t of its 256 chunks (Horner over KD_L coefficients each), then the scan
//             within the block: suf[i] = sum_{s >= i, s in block} c_s Z^(s-i); block aggregate S_B = suf[first].
//             Workgroup nblk: zpow[i] = Z^i, i < 256.  Workgroups after it: tpow[j] = (Z^256)^j, j < nblk.
//   kd_apply  workgroup B: the carry entering the block from above, G_B = sum_{B' > B} S_B' (Z^256)^(B'-B-1) (a direct sum over
//             the at most 1 024 aggregates, reduced through LDS); then K_t = suf[t+1] (same block) + Z^(last_in_block - t) G_B
//             and the replay of every chunk from its carry.
__global__ __launch_bounds__(KD_BLK) void kd_scan_kernel(KdPtrs a, KdStep pw, KdTop tw, uint32_t n, Fr* __restrict__ scratch) {
    __shared__ Fr sh[KD_BLK];
    const uint32_t KD_L = kd_len(n);
    const uint32_t m = (n + KD_L - 1) / KD_L, nblk = (m + KD_BLK - 1) / KD_BLK;
    Fr* suf = scratch + (size_t)blockIdx.y * kd_scratch_elems(n) + m;
    Fr* agg = suf + m;
    Fr* tpow = agg + nblk;
    Fr* zpow = tpow + nblk;
    if (blockIdx.x >= nblk) {
        // the tables of powers: Z^i (i < 256) by workgroup nblk, (Z^256)^j (j < nblk) by the ones after it
        const bool top = blockIdx.x > nblk;
        const uint32_t e0 = top ? (blockIdx.x - nblk - 1) * KD_BLK + threadIdx.x : threadIdx.x;
        if (top && e0 >= nblk) return;
        Fr acc = Fr::one();
        if (top) {
            for (uint32_t e = e0, j = 0; e; e >>= 1, j++)
                if (e & 1) acc = fe_mul(acc, tw.top[blockIdx.y][j]);  // (Z^256)^(2^j)
            fe_store(tpow + e0, acc);
        } else {
            for (uint32_t e = e0, j = 0; e; e >>= 1, j++)
                if (e & 1) acc = fe_mul(acc, pw.step[blockIdx.y][j]);  // Z^(2^j), j < 8
            fe_store(zpow + e0, acc);
        }
        return;
    }
    const uint32_t t = blockIdx.x * KD_BLK + threadIdx.x;
    Fr v = Fr::zero();
    {
        const uint32_t base = t * KD_L;
        if (base < n) {
            const Fr* __restrict__ p = a.p[blockIdx.y];
            const Fr z = a.z[blockIdx.y];
            const uint32_t top = min(n, base + KD_L);
            for (uint32_t i = top; i-- > base;) v = fe_add(fe_mul(v, z), fe_load(p + i));
        }
    }
    sh[threadIdx.x] = v;
    __syncthreads();
#pragma unroll 1
    for (uint32_t j = 0, d = 1; d < KD_BLK; d <<= 1, j++) {
        Fr o = Fr::zero();
        const bool has = threadIdx.x + d < KD_BLK;
        if (has) o = sh[threadIdx.x + d];
        __syncthreads();
        if (has) {
            v = fe_add(v, fe_mul(o, pw.step[blockIdx.y][j]));
            sh[threadIdx.x] = v;
        }
        __syncthreads();
    }
    if (t < m) fe_store(suf + t, v);
    if (threadIdx.x == 0) fe_store(agg + blockIdx.x, v);
}

__global__ __launch_bounds__(KD_BLK) void kd_apply_kernel(KdPtrs a, uint32_t n, Fr* __restrict__ scratch) {
    __shared__ Fr sh[KD_BLK];
    const uint32_t KD_L = kd_len(n);
    const uint32_t m = (n + KD_L - 1) / KD_L, nblk = (m + KD_BLK - 1) / KD_BLK;
    const Fr* suf = scratch + (size_t)blockIdx.y * kd_scratch_elems(n) + m;
    const Fr* agg = suf + m;
    const Fr* tpow = agg + nblk;
    const Fr* zpow = tpow + nblk;
    const uint32_t blk = blockIdx.x;
    // G = sum_{B' > blk} S_B' (Z^256)^(B' - blk - 1)
    Fr g = Fr::zero();
    for (uint32_t b = blk + 1 + threadIdx.x; b < nblk; b += KD_BLK) g = fe_add(g, fe_mul(fe_load(agg + b), fe_load(tpow + (b - blk - 1))));
    sh[threadIdx.x] = g;
    __syncthreads();
    for (uint32_t d = KD_BLK / 2; d; d >>= 1) {
        if (threadIdx.x < d) sh[threadIdx.x] = fe_add(sh[threadIdx.x], sh[threadIdx.x + d]);
        __syncthreads();
    }
    const Fr G = sh[0];
    const uint32_t t = blk * KD_BLK + threadIdx.x;
    const uint32_t base = t * KD_L;
    if (base >= n) return;
    const Fr* __restrict__ p = a.p[blockIdx.y];
    Fr* __restrict__ q = a.q[blockIdx.y];
    const Fr z = a.z[blockIdx.y];
    const uint32_t top = min(n, base + KD_L);
    // carry into chunk t: chunks above it in the same block + everything above the block
    const uint32_t last = min(m, (blk + 1) * KD_BLK) - 1;
    Fr run = (t < last) ? fe_load(suf + t + 1) : Fr::zero();
    run = fe_add(run, fe_mul(fe_load(zpow + (last - t)), G));
    // run = q[top - 1]; walk down: q[i-1] = p[i] + z q[i]  (p[i] is read before q[i] is written: q may be p)
    for (uint32_t i = top; i-- > base;) {
        const Fr pi = fe_load(p + i);
        fe_store(q + i, run);
        run = fe_add(fe_mul(run, z), pi);
    }
}

// `count` <= KD_MAX_BATCH divisions q[i] = (p[i] - p[i](z[i])) / (X - z[i]); scratch: count * kate_division_scratch(n)
void launch_kate_division_batch(const Fr* const* p, Fr* const* q, const Fr* z, uint32_t count, uint32_t n, Fr* scratch,
                                hipStream_t st) {
    const uint32_t KD_L = kd_len(n);
    const uint32_t m = (n + KD_L - 1) / KD_L;
    const uint32_t nblk = (m + KD_BLK - 1) / KD_BLK;
    KdPtrs pt;
    KdStep stp;
    KdTop tp;
    memset(&pt, 0, sizeof(pt));
    memset(&stp, 0, sizeof(stp));
    memset(&tp, 0, sizeof(tp));
    for (uint32_t b = 0; b < count; b++) {
        pt.p[b] = p[b];
        pt.q[b] = q[b];
        pt.z[b] = z[b];
        Fr Z = z[b];
        for (uint32_t i = 1; i < KD_L; i <<= 1) Z = fe_sqr(Z);  // z^KD_L
        Fr cur = Z;
        for (int j = 0; j < 8; j++) {
            stp.step[b][j] = cur;
            cur = fe_sqr(cur);
        }
        // cur = Z^256
        for (int j = 0; j < 10; j++) {
            tp.top[b][j] = cur;
            cur = fe_sqr(cur);
        }
    }
    hipLaunchKernelGGL(kd_scan_kernel, dim3(nblk + 1 + (nblk + KD_BLK - 1) / KD_BLK, count), dim3(KD_BLK), 0, st, pt, stp, tp, n, scratch);
    hipLaunchKernelGGL(kd_apply_kernel, dim3(nblk, count), dim3(KD_BLK), 0, st, pt, n, scratch);
}

void launch_kate_division(const Fr* p, Fr* q, uint32_t n, const Fr& z, Fr* scratch, hipStream_t st) {
    launch_kate_division_batch(&p, &q, &z, 1, n, scratch, st);
}

}  // namespace zk
