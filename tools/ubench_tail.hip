// ubench_tail.hip — what one general XYZZ addition on the carry-free field costs a LONE wave (the MSM reduction tails run
// at one or two waves per SIMD): pure register loop, with and without the 37-word shuffle of a tree step.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I webauthn-halo2_amd/csrc tools/ubench_tail.hip -o tools/ubench_tail
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include "ec29.hip.h"
using namespace zk;

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(64) void k_adds(const G1X29S* __restrict__ in, G1X29S* __restrict__ out, int iters) {
    const uint32_t t = blockIdx.x * 64 + threadIdx.x;
    G1X29 acc = g1x29_load(in + (t & 63));
    G1X29 b = g1x29_load(in + 64 + (t & 63));
#pragma unroll 1
    for (int i = 0; i < iters; i++) {
        G1X29 v = b;
        if (MODE == 1) v = g1x29_shfl_down(acc, 1 + (i & 31));
        g1x29_add(acc, v);
    }
    g1x29_store(out + t, acc);
}

int main() {
    // points: any field elements do for timing (the formulas do not test curve membership); limbs < 2^29, small top limb
    const int NP = 128;
    G1X29S* h = (G1X29S*)malloc(NP * sizeof(G1X29S));
    srand(7);
    for (int i = 0; i < NP; i++)
        for (int j = 0; j < 36; j++) h[i].w[j] = (j % 9 == 8) ? (rand() & 0xfffff) : (((uint32_t)rand() << 8 ^ rand()) & ((1u << 29) - 1));
    G1X29S *din, *dout;
    CHK(hipMalloc(&din, NP * sizeof(G1X29S)));
    CHK(hipMalloc(&dout, 1024 * 16 * 64 * sizeof(G1X29S)));
    CHK(hipMemcpy(din, h, NP * sizeof(G1X29S), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    const int iters = 64;
    for (int mode = 0; mode < 2; mode++)
        for (int waves : {64, 416, 1024, 4096, 16384}) {
            float best = 1e9;
            for (int rep = 0; rep < 5; rep++) {
                CHK(hipEventRecord(e0));
                if (mode == 0) hipLaunchKernelGGL(k_adds<0>, dim3(waves), dim3(64), 0, 0, din, dout, iters);
                else hipLaunchKernelGGL(k_adds<1>, dim3(waves), dim3(64), 0, 0, din, dout, iters);
                CHK(hipEventRecord(e1));
                CHK(hipEventSynchronize(e1));
                float ms;
                CHK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            printf("%s %6d waves x %d adds: %8.3f ms  -> %6.2f us per add per wave, %7.2f G adds/s\n", mode ? "shuffle+add" : "add        ", waves, iters,
                   best, best * 1e3 / iters, (double)waves * 64 * iters / (best * 1e-3) / 1e9);
        }
    return 0;
}
