// ubench_f29.hip — prototype of a carry-free 9 x 29-bit-limb Montgomery product (R = 2^261) against
// the 8 x 32-bit FIPS product of field.hip.h: throughput and a correctness cross-check.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I webauthn-halo2_amd/csrc tools/ubench_f29.hip -o tools/ubench_f29
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include "ec29.hip.h"
using namespace zk;

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

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
__device__ __forceinline__ Fe<PRM> loc_from29(const F29& a) {
    // limbs normalised (< 2^29), value < 2^256
    Fe<PRM> r;
#pragma unroll
    for (int w = 0; w < 8; w++) {
        // word w = bits [32w, 32w+32)
        const int i = (32 * w) / 29, o = 32 * w - 29 * i;  // starts in limb i at offset o
        uint64_t v = (uint64_t)a.l[i] >> o;
        v |= (uint64_t)a.l[i + 1] << (29 - o);
        if (i + 2 < 9) v |= (uint64_t)a.l[i + 2] << (58 - o);
        r.v[w] = (uint32_t)v;
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
__global__ __launch_bounds__(256) void mul32_kernel(Fe<PRM>* out, uint32_t seed, int iters) {
    Fe<PRM> x = Fe<PRM>::one(), y = Fe<PRM>::r2();
    x.v[0] += threadIdx.x + seed;
    y.v[1] ^= blockIdx.x;
    for (int i = 0; i < iters; i++) {
        x = fe_mul(x, y);
        y = fe_mul(y, x);
    }
    fe_store(out + blockIdx.x * 256 + threadIdx.x, fe_add(x, y));
}

// check: from29(mul29(to29(a), to29(b))) * 32 == fe_mul(a, b)  (mod p)
template <class PRM>
__global__ void check_kernel(uint32_t* bad, uint32_t seed) {
    Fe<PRM> a = Fe<PRM>::r2(), b = Fe<PRM>::one();
    a.v[0] ^= threadIdx.x * 2654435761u + seed;
    a.v[3] ^= blockIdx.x * 40503u;
    b.v[2] += threadIdx.x;
    a = fe_mul(a, a);
    b = fe_mul(b, a);  // two "random" canonical elements
    for (int it = 0; it < 8; it++) {
        const Fe<PRM> want = fe_mul(a, b);
        Fe<PRM> got = loc_from29<PRM>(loc_mul29<PRM>(loc_to29(a), loc_to29(b)));
        reduce_once(got);
        for (int k = 0; k < 5; k++) got = fe_add(got, got);
        bool eq = true;
        for (int k = 0; k < 8; k++) eq = eq && got.v[k] == want.v[k];
        if (!eq) atomicAdd(bad, 1u);
        a = fe_add(want, b);
        b = fe_mul(want, want);
    }
}

template <int MINW>
__global__ __launch_bounds__(64, MINW) void madd29_kernel(G1X* out, const G1Affine* pts, int iters) {
    G1X29 acc;
    acc.inf = true;
    const uint32_t t = blockIdx.x * 64 + threadIdx.x;
    for (int i = 0; i < iters; i++) {
        G1Affine p = affine_load(pts + ((t * 7 + i * 13) & 1023));
        if (!g1x29_add_affine(acc, p.x, p.y)) {
            G1X s = g1x29_to_std(acc);
            g1x_add_affine(s, p.x, p.y);
            acc = g1x29_from_std(s);
        }
    }
    g1x_store(out + t, g1x29_to_std(acc));
}
__global__ __launch_bounds__(64) void madd32_kernel(G1X* out, const G1Affine* pts, int iters) {
    G1X acc = G1X::identity();
    const uint32_t t = blockIdx.x * 64 + threadIdx.x;
    for (int i = 0; i < iters; i++) {
        G1Affine p = affine_load(pts + ((t * 7 + i * 13) & 1023));
        g1x_add_affine(acc, p.x, p.y);
    }
    g1x_store(out + t, acc);
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
    const int blocks = prop.multiProcessorCount * 8, mi = 512;
    uint32_t* bad; CHK(hipMalloc(&bad, 4)); CHK(hipMemset(bad, 0, 4));
    hipLaunchKernelGGL(check_kernel<FqParams>, dim3(64), dim3(256), 0, 0, bad, 7u);
    hipLaunchKernelGGL(check_kernel<FrParams>, dim3(64), dim3(256), 0, 0, bad, 9u);
    uint32_t hb = 1; CHK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
    printf("mismatches: %u\n", hb);
    Fq* fo; CHK(hipMalloc(&fo, (size_t)blocks * 256 * 32));
    float m1 = time_ms([&] { hipLaunchKernelGGL(mul32_kernel<FqParams>, dim3(blocks), dim3(256), 0, 0, fo, 1u, mi); });
    float m2 = time_ms([&] { hipLaunchKernelGGL(mul29_kernel<FqParams>, dim3(blocks), dim3(256), 0, 0, (uint32_t*)fo, 1u, mi); });
    double mm = (double)blocks * 256 * mi * 2;
    printf("fe_mul 8x32 FIPS asm  %8.3f ms  %.2f G modmul/s\n", m1, mm / m1 / 1e6);
    printf("mul29  9x29 carry-free %8.3f ms  %.2f G modmul/s\n", m2, mm / m2 / 1e6);
    // low occupancy (one wave per SIMD): the regime of the reduction tails
    const int lb = prop.multiProcessorCount;  // 256 threads = 4 waves per CU
    float m3 = time_ms([&] { hipLaunchKernelGGL(mul32_kernel<FqParams>, dim3(lb), dim3(256), 0, 0, fo, 1u, mi); });
    float m4 = time_ms([&] { hipLaunchKernelGGL(mul29_kernel<FqParams>, dim3(lb), dim3(256), 0, 0, (uint32_t*)fo, 1u, mi); });
    printf("one wave per SIMD: 8x32 %.3f ms (%.0f cycles/product), 9x29 %.3f ms (%.0f cycles/product)\n", m3,
           m3 * 1e-3 * prop.clockRate * 1e3 / (mi * 2), m4, m4 * 1e-3 * prop.clockRate * 1e3 / (mi * 2));
    // XYZZ mixed addition: points = small multiples of the generator
    static G1Affine hp[1024];
    {
        G1X cur = G1X::identity();
        Fq gx = Fq::one(), gy = fe_add(Fq::one(), Fq::one());
        for (int i = 0; i < 1024; i++) {
            g1x_add_affine(cur, gx, gy);
            Fq t = fe_inv(cur.zzz), u = fe_mul(cur.zz, t);
            hp[i].x = fe_mul(cur.x, fe_sqr(u));
            hp[i].y = fe_mul(cur.y, t);
        }
    }
    G1Affine* dp; CHK(hipMalloc(&dp, sizeof(hp))); CHK(hipMemcpy(dp, hp, sizeof(hp), hipMemcpyHostToDevice));
    const int ab = prop.multiProcessorCount * 4 * 12, ai = 128;
    G1X* ao; CHK(hipMalloc(&ao, (size_t)ab * 64 * sizeof(G1X)));
    G1X* ao2; CHK(hipMalloc(&ao2, (size_t)ab * 64 * sizeof(G1X)));
    float a0 = time_ms([&] { hipLaunchKernelGGL(madd32_kernel, dim3(ab), dim3(64), 0, 0, ao, dp, ai); });
    float a1 = time_ms([&] { hipLaunchKernelGGL(madd29_kernel<1>, dim3(ab), dim3(64), 0, 0, ao2, dp, ai); });
    float a2 = time_ms([&] { hipLaunchKernelGGL(madd29_kernel<4>, dim3(ab), dim3(64), 0, 0, ao2, dp, ai); });
    float a3 = time_ms([&] { hipLaunchKernelGGL(madd29_kernel<5>, dim3(ab), dim3(64), 0, 0, ao2, dp, ai); });
    const double na = (double)ab * 64 * ai;
    printf("xyzz mixed add 8x32: %.2f G/s;  9x29: %.2f G/s (no bound), %.2f (4 waves/SIMD), %.2f (5 waves/SIMD)\n", na / a0 / 1e6,
           na / a1 / 1e6, na / a2 / 1e6, na / a3 / 1e6);
    // same result?
    static G1X h0[64], h1[64];
    CHK(hipMemcpy(h0, ao, sizeof(h0), hipMemcpyDeviceToHost));
    CHK(hipMemcpy(h1, ao2, sizeof(h1), hipMemcpyDeviceToHost));
    int diff = 0;
    for (int i = 0; i < 64; i++) {
        // compare affine x: X / ZZ
        Fq a = fe_mul(h0[i].x, h1[i].zz), b = fe_mul(h1[i].x, h0[i].zz);
        for (int k = 0; k < 8; k++) diff += a.v[k] != b.v[k];
    }
    printf("mixed-add cross-check differences: %d\n", diff);
    return 0;
}
