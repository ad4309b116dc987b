// ubench_f64.hip — the double-precision route for the 256-bit Montgomery product, measured against the 9 x 29-bit integer
// product the engine uses (field29.hip.h; the copy below is the plain form of tools/ubench_f29.hip).
//
//   A. exact small limbs: 11 limbs of 24 bits in doubles, R = 2^264.  Every partial product is ONE v_fma_f64 that also
//      accumulates (a_i b_j + t, 48-bit products, at most 22 of them per column: < 2^53, exact) — the floating-point twin of
//      v_mad_u64_u32 — so the product is 121 FMAs, the interleaved Montgomery reduction 121 more plus, per round, the
//      "mod 2^24" steps that integer code gets from a mask (v_mul / v_trunc / v_fma triples), and a carry pass at the end:
//      374 f64 instructions against 258 integer ones.
//   B. (counted, not built) 52-bit limbs with hi / lo FMA pairs in round-toward-zero (5 limbs: 25 + 25 partial products x
//      [2 FMAs + 1 subtraction + 2 64-bit integer additions of the raw patterns]) ~ 290 instructions of the 4.6 - 4.8-cycle
//      class: between A and the integer product — see the rates printed below and DESIGN.md §4.
//
// Both products are checked against fe_mul (8 x 32 FIPS) on the same operands.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I webauthn-halo2_amd/csrc tools/ubench_f64.hip -o tools/ubench_f64
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include "ec29.hip.h"
using namespace zk;

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// ------------------------------------------------------------------ integer: 9 x 29 ---
template <class PRM>
struct P29L {
    static constexpr uint32_t limb(int i) {
        const int bit = 29 * i, w = bit >> 5, o = bit & 31;
        uint64_t two = PRM::P[w];
        if (w + 1 < 8) two |= (uint64_t)PRM::P[w + 1] << 32;
        return (uint32_t)(two >> o) & M29;
    }
    static constexpr uint32_t INV = PRM::INV & M29;
};
struct F29 {
    uint32_t l[9];
};
template <class PRM>
__device__ __forceinline__ F29 loc_to29(const Fe<PRM>& a) {
    F29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int bit = 29 * i, w = bit >> 5, o = bit & 31;
        uint64_t two = a.v[w];
        if (w + 1 < 8) two |= (uint64_t)a.v[w + 1] << 32;
        r.l[i] = (uint32_t)(two >> o) & M29;
    }
    return r;
}
template <class PRM>
__device__ __forceinline__ F29 loc_mul29(const F29& a, const F29& b) {
    uint32_t m[9];
    F29 r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * P29L<PRM>::limb(k - i);
        m[k] = ((uint32_t)acc * P29L<PRM>::INV) & M29;
        acc += (uint64_t)m[k] * P29L<PRM>::limb(0);
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
#pragma unroll
        for (int i = k - 8; i < 9; i++) acc += (uint64_t)a.l[i] * b.l[k - i];
#pragma unroll
        for (int i = k - 8; i < 9; i++) acc += (uint64_t)m[i] * P29L<PRM>::limb(k - i);
        r.l[k - 9] = (uint32_t)acc & M29;
        acc >>= 29;
    }
    r.l[8] = (uint32_t)acc;
    return r;
}

// ------------------------------------------------------------ double: 11 x 24, R = 2^264 ---
template <class PRM>
struct P24L {
    static constexpr uint32_t ilimb(int i) {
        const int bit = 24 * i, w = bit >> 5, o = bit & 31;
        if (w >= 8) return 0;
        uint64_t two = PRM::P[w];
        if (w + 1 < 8) two |= (uint64_t)PRM::P[w + 1] << 32;
        return (uint32_t)(two >> o) & 0xffffffu;
    }
    static constexpr double limb(int i) { return (double)ilimb(i); }
    static constexpr double INV = (double)(PRM::INV & 0xffffffu);  // -p^-1 mod 2^24
};
struct F64 {
    double l[11];
};
template <class PRM>
__device__ __forceinline__ F64 to64(const Fe<PRM>& a) {
    F64 r;
#pragma unroll
    for (int i = 0; i < 11; i++) {
        const int bit = 24 * i, w = bit >> 5, o = bit & 31;
        uint64_t two = w < 8 ? a.v[w] : 0u;
        if (w + 1 < 8) two |= (uint64_t)a.v[w + 1] << 32;
        r.l[i] = (double)((uint32_t)(two >> o) & 0xffffffu);
    }
    return r;
}
template <class PRM>
__device__ __forceinline__ Fe<PRM> from64(const F64& a) {  // limbs < 2^24 (the top one whatever is left), value < 2^256
    Fe<PRM> r;
    uint64_t acc = 0;
    int bits = 0, w = 0;
#pragma unroll
    for (int i = 0; i < 11; i++) {
        acc |= (uint64_t)(uint32_t)a.l[i] << bits;
        bits += 24;
        if (bits >= 32 && w < 8) {
            r.v[w++] = (uint32_t)acc;
            acc >>= 32;
            bits -= 32;
        }
    }
    return r;
}
// a, b: limbs in [0, 2^24), values < 2p  ->  a b 2^-264 mod p, < 2p, limbs in [0, 2^24)
template <class PRM>
__device__ __forceinline__ F64 mul64(const F64& a, const F64& b) {
    double t[22];
#pragma unroll
    for (int k = 0; k < 22; k++) t[k] = 0.0;
#pragma unroll
    for (int i = 0; i < 11; i++)
#pragma unroll
        for (int j = 0; j < 11; j++) t[i + j] = __builtin_fma(a.l[i], b.l[j], t[i + j]);
#pragma unroll
    for (int i = 0; i < 11; i++) {
        const double q = __builtin_trunc(t[i] * 0x1p-24);
        const double lo = __builtin_fma(q, -0x1p24, t[i]);  // t[i] mod 2^24
        const double pr = lo * P24L<PRM>::INV;              // < 2^48
        const double q2 = __builtin_trunc(pr * 0x1p-24);
        const double m = __builtin_fma(q2, -0x1p24, pr);    // (t[i] p') mod 2^24
#pragma unroll
        for (int j = 0; j < 11; j++) t[i + j] = __builtin_fma(m, P24L<PRM>::limb(j), t[i + j]);
        t[i + 1] = __builtin_fma(t[i], 0x1p-24, t[i + 1]);  // t[i] is a multiple of 2^24 now: exact
    }
    F64 r;
    double carry = 0.0;
#pragma unroll
    for (int k = 0; k < 10; k++) {
        const double v = t[11 + k] + carry;
        carry = __builtin_trunc(v * 0x1p-24);
        r.l[k] = __builtin_fma(carry, -0x1p24, v);
    }
    r.l[10] = t[21] + carry;
    return r;
}

template <class PRM>
__global__ __launch_bounds__(256) void mul29_kernel(uint32_t* out, uint32_t seed, int iters) {
    Fe<PRM> x0 = Fe<PRM>::one(), y0 = Fe<PRM>::r2();
    x0.v[0] += threadIdx.x + seed;
    y0.v[1] ^= blockIdx.x;
    F29 x = loc_to29(x0), y = loc_to29(y0);
    for (int i = 0; i < iters; i++) {
        x = loc_mul29<PRM>(x, y);
        y = loc_mul29<PRM>(y, x);
    }
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) s += x.l[i] ^ y.l[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <class PRM>
__global__ __launch_bounds__(256) void mul64_kernel(uint32_t* out, uint32_t seed, int iters) {
    Fe<PRM> x0 = Fe<PRM>::one(), y0 = Fe<PRM>::r2();
    x0.v[0] += threadIdx.x + seed;
    y0.v[1] ^= blockIdx.x;
    F64 x = to64(x0), y = to64(y0);
    for (int i = 0; i < iters; i++) {
        x = mul64<PRM>(x, y);
        y = mul64<PRM>(y, x);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 11; i++) s += x.l[i] - y.l[i];
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(int64_t)s;
}

// from64(mul64(a, b)) * 2^8 == fe_mul(a, b) (mod p); and with a lazily reduced operand (the product itself, < 2p):
// from64(mul64(mul64(a, b), b)) * 2^16 == fe_mul(fe_mul(a, b), b)
template <class PRM>
__device__ __forceinline__ bool same(const F64& g, int doublings, const Fe<PRM>& want) {
    Fe<PRM> got = from64<PRM>(g);
    reduce_once(got);
    for (int k = 0; k < doublings; k++) got = fe_add(got, got);
    bool eq = true;
    for (int k = 0; k < 8; k++) eq = eq && got.v[k] == want.v[k];
    return eq;
}
template <class PRM>
__global__ void check_kernel(uint32_t* bad, uint32_t seed) {
    Fe<PRM> a = Fe<PRM>::r2(), b = Fe<PRM>::one();
    a.v[0] ^= threadIdx.x * 2654435761u + seed;
    a.v[3] ^= blockIdx.x * 40503u;
    b.v[2] += threadIdx.x;
    a = fe_mul(a, a);
    b = fe_mul(b, a);  // two "random" canonical elements
    for (int it = 0; it < 8; it++) {
        const Fe<PRM> w1 = fe_mul(a, b), w2 = fe_mul(w1, b);
        const F64 xb = to64(b);
        const F64 g = mul64<PRM>(to64(a), xb);
        const F64 h = mul64<PRM>(g, xb);
        if (!same<PRM>(g, 8, w1) || !same<PRM>(h, 16, w2)) atomicAdd(bad, 1u);
        a = fe_add(w1, b);
        b = fe_mul(w2, w1);
    }
}

template <class F>
float time_ms(F f) {
    hipEvent_t a, b;
    CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    f();
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(a));
    f();
    CHK(hipEventRecord(b));
    CHK(hipEventSynchronize(b));
    float ms; CHK(hipEventElapsedTime(&ms, a, b));
    return ms;
}

int main() {
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    uint32_t* bad; CHK(hipMalloc(&bad, 4)); CHK(hipMemset(bad, 0, 4));
    hipLaunchKernelGGL(check_kernel<FqParams>, dim3(64), dim3(256), 0, 0, bad, 7u);
    hipLaunchKernelGGL(check_kernel<FrParams>, dim3(64), dim3(256), 0, 0, bad, 9u);
    uint32_t hb = 1; CHK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
    printf("f64 product against fe_mul, 2 fields x 16384 lanes x 8 pairs, canonical and lazily reduced operands: %u mismatches\n", hb);
    const int mi = 512;
    uint32_t* fo; CHK(hipMalloc(&fo, (size_t)prop.multiProcessorCount * 8 * 256 * 4));
    for (int per_cu : {8, 4, 1}) {  // workgroups of 256 lanes per CU: 8 / 4 / 1 waves per SIMD
        const int blocks = prop.multiProcessorCount * per_cu;
        const float m29 = time_ms([&] { hipLaunchKernelGGL(mul29_kernel<FqParams>, dim3(blocks), dim3(256), 0, 0, fo, 1u, mi); });
        const float m64 = time_ms([&] { hipLaunchKernelGGL(mul64_kernel<FqParams>, dim3(blocks), dim3(256), 0, 0, fo, 1u, mi); });
        const double mm = (double)blocks * 256 * mi * 2;
        // cycles of one SIMD per wave-product: waves per SIMD x time x clock / products per wave
        const double cyc29 = m29 * 1e-3 * prop.clockRate * 1e3 / (mi * 2) / per_cu;
        const double cyc64 = m64 * 1e-3 * prop.clockRate * 1e3 / (mi * 2) / per_cu;
        printf("%d waves/SIMD: 9x29 integer %7.3f ms = %6.2f G products/s (%5.0f SIMD cycles per wave-product); 11x24 f64 %7.3f ms = %6.2f G products/s (%5.0f) -> f64 / integer = %.2fx the time\n",
               per_cu, m29, mm / m29 / 1e6, cyc29, m64, mm / m64 / 1e6, cyc64, m64 / m29);
    }
    return 0;
}
