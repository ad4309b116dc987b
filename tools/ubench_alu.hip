// ubench_alu.hip — integer-ALU microbenchmarks that size the field-arithmetic
// design on gfx950: rate of v_mad_u64_u32 / v_mul_lo / v_mul_hi / v_fma_f64 and
// the throughput of the 8x32-limb Montgomery product and the XYZZ mixed add.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I webauthn-halo2_amd/csrc tools/ubench_alu.hip -o tools/ubench_alu
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include "ec.hip.h"
#include "field.hip.h"
using namespace zk;

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int OP>
__global__ __launch_bounds__(256) void alu_kernel(uint64_t* out, uint32_t seed, int iters) {
    uint32_t a = seed + threadIdx.x, b = seed * 3 + blockIdx.x;
    uint64_t acc[8];
    double d[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { acc[k] = a + k; d[k] = (double)(a + k); }
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int k = 0; k < 8; k++) {
            if (OP == 0) acc[k] = (uint64_t)(uint32_t)acc[k] * b + acc[k];              // v_mad_u64_u32
            if (OP == 1) acc[k] = (uint32_t)((uint32_t)acc[k] * b) + (uint32_t)k;        // v_mul_lo_u32
            if (OP == 2) acc[k] = __umulhi((uint32_t)acc[k], b) + a;                     // v_mul_hi_u32
            if (OP == 3) d[k] = fma(d[k], 1.0000001, 0.5);                               // v_fma_f64
            if (OP == 4) acc[k] = (uint32_t)acc[k] + b;                                  // v_add_u32
            if (OP == 5) acc[k] = __umul24((uint32_t)acc[k], b) + k;                     // v_mad_u32_u24
        }
    }
    uint64_t r = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) r += acc[k] + (uint64_t)d[k];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <class PRM>
__global__ __launch_bounds__(256) void modmul_kernel(Fe<PRM>* out, uint32_t seed, int iters) {
    Fe<PRM> x = Fe<PRM>::one(), y = Fe<PRM>::r2();
    x.v[0] += threadIdx.x + seed;
    y.v[1] ^= blockIdx.x;
    for (int i = 0; i < iters; i++) {
        x = fe_mul(x, y);
        y = fe_mul(y, x);
    }
    fe_store(out + blockIdx.x * 256 + threadIdx.x, fe_add(x, y));
}

__global__ __launch_bounds__(64) void madd_kernel(G1X* out, const G1Affine* pts, int iters) {
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
    f();  // warm
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
    printf("device %s CUs %d clock %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
    const int blocks = prop.multiProcessorCount * 8, iters = 4096;
    uint64_t* out; CHK(hipMalloc(&out, (size_t)blocks * 256 * 8));
    const char* names[6] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_fma_f64", "v_add_u32", "v_mad_u32_u24"};
    float ms[6];
    ms[0] = time_ms([&] { hipLaunchKernelGGL(alu_kernel<0>, dim3(blocks), dim3(256), 0, 0, out, 1u, iters); });
    ms[1] = time_ms([&] { hipLaunchKernelGGL(alu_kernel<1>, dim3(blocks), dim3(256), 0, 0, out, 1u, iters); });
    ms[2] = time_ms([&] { hipLaunchKernelGGL(alu_kernel<2>, dim3(blocks), dim3(256), 0, 0, out, 1u, iters); });
    ms[3] = time_ms([&] { hipLaunchKernelGGL(alu_kernel<3>, dim3(blocks), dim3(256), 0, 0, out, 1u, iters); });
    ms[4] = time_ms([&] { hipLaunchKernelGGL(alu_kernel<4>, dim3(blocks), dim3(256), 0, 0, out, 1u, iters); });
    ms[5] = time_ms([&] { hipLaunchKernelGGL(alu_kernel<5>, dim3(blocks), dim3(256), 0, 0, out, 1u, iters); });
    for (int i = 0; i < 6; i++) {
        double ops = (double)blocks * 256 * iters * 8;
        printf("%-16s %8.3f ms  %8.2f Tops/s  (%.2f lane-ops/clk/CU at %.1f GHz)\n", names[i], ms[i], ops / ms[i] / 1e9,
               ops / (ms[i] * 1e-3) / prop.multiProcessorCount / (prop.clockRate * 1e3), prop.clockRate / 1e6);
    }
    Fr* fo; CHK(hipMalloc(&fo, (size_t)blocks * 256 * 32));
    const int mi = 512;
    float m1 = time_ms([&] { hipLaunchKernelGGL(modmul_kernel<FrParams>, dim3(blocks), dim3(256), 0, 0, fo, 1u, mi); });
    float m2 = time_ms([&] { hipLaunchKernelGGL(modmul_kernel<FqParams>, dim3(blocks), dim3(256), 0, 0, (Fq*)fo, 1u, mi); });
    double mm = (double)blocks * 256 * mi * 2;
    printf("fe_mul<Fr> %8.3f ms  %.2f G modmul/s\n", m1, mm / m1 / 1e6);
    printf("fe_mul<Fq> %8.3f ms  %.2f G modmul/s\n", m2, mm / m2 / 1e6);
    // mixed add throughput: points = small multiples of the generator built on host
    G1Affine hp[1024];
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
    const int ab = prop.multiProcessorCount * 16, ai = 256;
    G1X* ao; CHK(hipMalloc(&ao, (size_t)ab * 64 * sizeof(G1X)));
    float m3 = time_ms([&] { hipLaunchKernelGGL(madd_kernel, dim3(ab), dim3(64), 0, 0, ao, dp, ai); });
    printf("xyzz mixed add %8.3f ms  %.2f G add/s (%d waves)\n", m3, (double)ab * 64 * ai / m3 / 1e6, ab);
    const int ab2 = prop.multiProcessorCount * 4 * 3;
    float m4 = time_ms([&] { hipLaunchKernelGGL(madd_kernel, dim3(ab2), dim3(64), 0, 0, ao, dp, ai); });
    printf("xyzz mixed add %8.3f ms  %.2f G add/s (%d waves = 3/SIMD)\n", m4, (double)ab2 * 64 * ai / m4 / 1e6, ab2);
    return 0;
}
