// ubench_bitsum.hip — the MSM bit-sum kernel alone on synthetic bucket sums, with wall-clock stamps of workgroup 0
// (serial loads, wave tree, barrier, cross-wave tree).  Build:
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -DZK_TAIL_TRACE -I webauthn-halo2_amd/csrc tools/ubench_bitsum.hip -o tools/ubench_bitsum
#include "msm.hip"

#include <stdio.h>
#include <stdlib.h>
using namespace zk;

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(64) void k_heavy(const G1X29S* __restrict__ in, G1X29S* __restrict__ out, int iters) {
    const uint32_t t = blockIdx.x * 64 + threadIdx.x;
    G1X29 acc = g1x29_load(in + (t & 63) * 8);
    const G1X29 b = g1x29_load(in + (64 + (t & 63)) * 8);
#pragma unroll 1
    for (int i = 0; i < iters; i++) g1x29_add(acc, b);
    if (acc.inf) g1x29_store(out + t, acc);
}

int main() {
    const uint32_t c = 13, nb = 1u << (c - 1), parts = 8, split = 4;
    const size_t np = (size_t)nb * parts;
    G1X29S* h = (G1X29S*)calloc(np, sizeof(G1X29S));
    srand(7);
    for (size_t b = 0; b < nb; b++)
        for (int j = 0; j < 36; j++) h[b * parts].w[j] = (j % 9 == 8) ? (rand() & 0xfffff) : (((uint32_t)rand() << 8 ^ rand()) & ((1u << 29) - 1));
    uint32_t* hs = (uint32_t*)malloc((nb + 1) * 4);
    for (uint32_t b = 0; b <= nb; b++) hs[b] = b * 40 * PAD;
    G1X29S* dpart;
    uint32_t* dstart;
    G1X* dout;
    CHK(hipMalloc(&dpart, np * sizeof(G1X29S)));
    CHK(hipMalloc(&dstart, (nb + 1) * 4));
    CHK(hipMalloc(&dout, c * split * sizeof(G1X)));
    CHK(hipMemcpy(dpart, h, np * sizeof(G1X29S), hipMemcpyHostToDevice));
    CHK(hipMemcpy(dstart, hs, (nb + 1) * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    CHK(hipFree(dout));
    CHK(hipMalloc(&dout, c * 64 * sizeof(G1X)));
    for (int cfg = 0; cfg < 7; cfg++) {
        const uint32_t threads[7] = {512, 256, 256, 128, 128, 64, 64}, splits[7] = {4, 4, 8, 4, 16, 4, 32};
        for (int rep = 0; rep < 3; rep++) {
            const uint32_t sp = splits[cfg];
            CHK(hipEventRecord(e0));
            if (threads[cfg] == 512) hipLaunchKernelGGL(msm_bitsum_kernel<512>, dim3(c * sp), dim3(512), 0, 0, dpart, parts, nb, c, sp, dstart, dout);
            if (threads[cfg] == 256) hipLaunchKernelGGL(msm_bitsum_kernel<256>, dim3(c * sp), dim3(256), 0, 0, dpart, parts, nb, c, sp, dstart, dout);
            if (threads[cfg] == 128) hipLaunchKernelGGL(msm_bitsum_kernel<128>, dim3(c * sp), dim3(128), 0, 0, dpart, parts, nb, c, sp, dstart, dout);
            if (threads[cfg] == 64) hipLaunchKernelGGL(msm_bitsum_kernel<64>, dim3(c * sp), dim3(64), 0, 0, dpart, parts, nb, c, sp, dstart, dout);
            CHK(hipEventRecord(e1));
            CHK(hipEventSynchronize(e1));
            float ms;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long tr[16];
            CHK(hipMemcpyFromSymbol(tr, HIP_SYMBOL(zk_tail_trace), sizeof(tr)));
            if (rep) printf("threads %3u split %2u: kernel %.1f us | workgroup 0: serial %.1f, wave tree %.1f, barrier %.1f, cross-wave tree %.1f us\n", threads[cfg], sp, ms * 1e3,
                   (tr[2] - tr[0]) * 0.01, (tr[3] - tr[2]) * 0.01, threads[cfg] > 64 ? (tr[4] - tr[3]) * 0.01 : 0.0, threads[cfg] > 64 ? (tr[5] - tr[4]) * 0.01 : 0.0);
        }
    }
    {
        // second-level gather: 4096 buckets of 40 first-level partials each, 8 parts (only part 0 used), 16-lane groups
        const uint32_t per = 40;
        G1X29S* dpartial;
        CHK(hipMalloc(&dpartial, (size_t)nb * (per + 8) * sizeof(G1X29S)));
        for (uint32_t b = 0; b < per; b++) CHK(hipMemcpy(dpartial + (size_t)b * nb, dpart, 0, hipMemcpyDeviceToDevice));
        G1X29S* hp = (G1X29S*)malloc((size_t)nb * (per + 8) * sizeof(G1X29S));
        for (size_t i = 0; i < (size_t)nb * (per + 8); i++) hp[i] = h[(i % nb) * parts];
        CHK(hipMemcpy(dpartial, hp, (size_t)nb * (per + 8) * sizeof(G1X29S), hipMemcpyHostToDevice));
        const uint32_t ngroups = nb * parts;
        if (getenv("RAGGED")) {  // bucket sizes 37 .. 44 partials instead of 40 each
            uint32_t o = 0;
            for (uint32_t b = 0; b <= nb; b++) { hs[b] = o * PAD; o += 37 + rand() % 8; }
            CHK(hipMemcpy(dstart, hs, (nb + 1) * 4, hipMemcpyHostToDevice));
        }
        for (int rep = 0; rep < 6; rep++) {
            if (rep >= 3) {  // right after a chip-filling kernel (as in the prover: the accumulation precedes the tail)
                hipLaunchKernelGGL(k_heavy, dim3(16384), dim3(64), 0, 0, dpart, dpartial, 16);
                printf("after a heavy kernel: ");
            }
            CHK(hipEventRecord(e0));
            hipLaunchKernelGGL(msm_gather_kernel<16>, dim3((ngroups * 16 + 255) / 256), dim3(256), 0, 0, dstart, dpartial, parts, ngroups, dpart);
            CHK(hipEventRecord(e1));
            CHK(hipEventSynchronize(e1));
            float ms;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            unsigned long long tr[16];
            CHK(hipMemcpyFromSymbol(tr, HIP_SYMBOL(zk_tail_trace), sizeof(tr)));
            printf("gather<16>: kernel %.1f us | workgroup 0: serial %.1f, tree %.1f us\n", ms * 1e3, (tr[9] - tr[8]) * 0.01, (tr[10] - tr[9]) * 0.01);
        }
    }
    return 0;
}
