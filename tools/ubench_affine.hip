// ubench_affine.hip — settles DESIGN.md §4 "batch-affine" by measurement: bucket additions as BATCHED AFFINE
// additions (Montgomery's trick: one field inversion per batch) against the XYZZ mixed addition the accumulation runs, in the
// same field arithmetic, same launch shape (64-lane workgroups, 4 waves per SIMD), same operands (gathers from a point table).
//
//   xyzz      one running sum per lane, `iters` dependent mixed additions (8M + 2S, no inversion): the accumulation's inner loop
//   affine<B> B independent running sums per lane (affine, kept in memory: B x 64 bytes per lane do not fit the registers), one
//             lane-local inversion per round of B additions: forward pass (dx_i, prefix products), fe_inv (Fermat, ~ 254 S + 127 M),
//             backward pass (2 products for the individual inverse, lambda, lambda^2, y3): 5M + 1S per addition + inversion / B.
//             Running B = 16, 64, 256 separates the two costs: t(B) = a + I / B -> a (the "free inversion" floor) and I.
// Both kernels produce the same sums (checked: lane sums of the affine kernel equal the XYZZ kernel's, normalised on the host).
// The arithmetic is the 8 x 32-bit Montgomery product of field.hip.h for BOTH kernels (the ratio is what matters; the accumulation's
// 9 x 29-bit form speeds both sides alike: its products cost 0.72 of these).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I webauthn-halo2_amd/csrc tools/ubench_affine.hip -o tools/ubench_affine
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "ec.hip.h"
#include "hostutil.h"
using namespace zk;

#define CHK(x)                                                                 \
    do {                                                                       \
        hipError_t e = (x);                                                    \
        if (e != hipSuccess) {                                                 \
            printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__);    \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

static constexpr uint32_t TABLE = 4096;  // points of the gather table (256 KiB: L2-resident; the real table is 512 MiB)

__device__ __forceinline__ uint32_t pick(uint32_t t, uint32_t chain, uint32_t step) { return (t * 2654435761u + chain * 40503u + step * 97u) & (TABLE - 1); }

// table[j] = s_j * G, s_j a 62-bit pseudo-random scalar: sums of a few hundred of them never meet a doubling or a cancellation
__global__ void table_kernel(G1Affine* table) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= TABLE) return;
    uint64_t s = (0x9E3779B97F4A7C15ull * (j + 1)) ^ (0xD1B54A32D192ED03ull * (j + 7));
    s |= 1ull << 61;
    Fq gx = Fq::one(), gy = fe_add(Fq::one(), Fq::one());  // G = (1, 2)
    G1X acc = G1X::identity();
    for (int b = 61; b >= 0; b--) {
        if (!acc.is_identity()) acc = g1x_dbl(acc);
        if ((s >> b) & 1) g1x_add_affine(acc, gx, gy);
    }
    const Fq t = fe_inv(acc.zzz), u = fe_mul(acc.zz, t);
    G1Affine r;
    r.x = fe_mul(acc.x, fe_sqr(u));
    r.y = fe_mul(acc.y, t);
    fe_store(&table[j].x, r.x);
    fe_store(&table[j].y, r.y);
}

// chain c of lane t: start = table[pick(t, c, 0)], then + table[pick(t, c, s)] for s = 1 .. iters
template <int MINW>
__global__ __launch_bounds__(64, MINW) void xyzz_kernel(const G1Affine* __restrict__ table, G1X* __restrict__ out, uint32_t chains, uint32_t iters) {
    const uint32_t t = blockIdx.x * 64 + threadIdx.x, nt = gridDim.x * 64;
    for (uint32_t c = 0; c < chains; c++) {
        G1X acc = G1X::identity();
        for (uint32_t s = 0; s <= iters; s++) {
            const G1Affine p = affine_load(table + pick(t, c, s));
            g1x_add_affine(acc, p.x, p.y);
        }
        g1x_store(out + (size_t)c * nt + t, acc);
    }
}

template <int B, int MINW>
__global__ __launch_bounds__(64, MINW) void affine_kernel(const G1Affine* __restrict__ table, G1Affine* __restrict__ sums, Fq* __restrict__ pre,
                                                          uint32_t iters) {
    const uint32_t t = blockIdx.x * 64 + threadIdx.x, nt = gridDim.x * 64;
    for (uint32_t c = 0; c < B; c++) {  // the running sums start as the chains' first points
        const G1Affine p = affine_load(table + pick(t, c, 0));
        fe_store(&sums[(size_t)c * nt + t].x, p.x);
        fe_store(&sums[(size_t)c * nt + t].y, p.y);
    }
    for (uint32_t s = 1; s <= iters; s++) {
        // forward: prefix products of the dx_i = x2 - x1 (one product per addition)
        Fq run = Fq::one();
        for (uint32_t c = 0; c < B; c++) {
            const Fq x1 = fe_load(&sums[(size_t)c * nt + t].x);
            const Fq x2 = fe_load(&table[pick(t, c, s)].x);
            fe_store(pre + (size_t)c * nt + t, run);
            run = fe_mul(run, fe_sub(x2, x1));
        }
        Fq inv = fe_inv(run);  // one inversion per round of B additions
        // backward: individual inverses (two products), lambda, x3, y3 (two products, one squaring)
        for (uint32_t c = B; c-- > 0;) {
            const G1Affine a = affine_load(sums + (size_t)c * nt + t);
            const G1Affine p = affine_load(table + pick(t, c, s));
            const Fq dx = fe_sub(p.x, a.x);
            const Fq di = fe_mul(inv, fe_load(pre + (size_t)c * nt + t));
            inv = fe_mul(inv, dx);
            const Fq lam = fe_mul(fe_sub(p.y, a.y), di);
            const Fq x3 = fe_sub(fe_sub(fe_sqr(lam), a.x), p.x);
            const Fq y3 = fe_sub(fe_mul(lam, fe_sub(a.x, x3)), a.y);
            fe_store(&sums[(size_t)c * nt + t].x, x3);
            fe_store(&sums[(size_t)c * nt + t].y, y3);
        }
    }
}

static G1Affine norm_host(const G1X& p) {
    G1Affine r;
    const Fq t = fe_inv(p.zzz), u = fe_mul(p.zz, t);
    r.x = fe_mul(p.x, fe_sqr(u));
    r.y = fe_mul(p.y, t);
    return r;
}

template <int B>
static double run_affine(const G1Affine* table, uint32_t waves, uint32_t iters, std::vector<G1Affine>* out) {
    const uint32_t nt = waves * 64;
    G1Affine* sums;
    Fq* pre;
    CHK(hipMalloc(&sums, (size_t)B * nt * sizeof(G1Affine)));
    CHK(hipMalloc(&pre, (size_t)B * nt * sizeof(Fq)));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL((affine_kernel<B, 4>), dim3(waves), dim3(64), 0, 0, table, sums, pre, iters);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((affine_kernel<B, 4>), dim3(waves), dim3(64), 0, 0, table, sums, pre, iters);
    CHK(hipEventRecord(e1, 0));
    CHK(hipDeviceSynchronize());
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    if (out) {
        out->resize((size_t)B * nt);
        CHK(hipMemcpy(out->data(), sums, out->size() * sizeof(G1Affine), hipMemcpyDeviceToHost));
    }
    CHK(hipFree(sums));
    CHK(hipFree(pre));
    return ms;
}

int main() {
    G1Affine* table;
    CHK(hipMalloc(&table, TABLE * sizeof(G1Affine)));
    hipLaunchKernelGGL(table_kernel, dim3(TABLE / 64), dim3(64), 0, 0, table);
    CHK(hipDeviceSynchronize());
    const uint32_t waves = 256 * 4 * 4;  // 4 waves per SIMD on 1024 SIMDs: one resident round, as the accumulation is launched
    const uint32_t nt = waves * 64, iters = 16;
    const double clk = 2.1e9;  // the clock the chip sustains under the accumulation (profiles/r4_pmc_ops.txt)
    // ---- XYZZ: 16 chains of 16 additions per lane (so that both kernels do the same 256 additions per lane at B = 16)
    G1X* xo;
    CHK(hipMalloc(&xo, (size_t)16 * nt * sizeof(G1X)));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL((xyzz_kernel<4>), dim3(waves), dim3(64), 0, 0, table, xo, 16u, iters);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((xyzz_kernel<4>), dim3(waves), dim3(64), 0, 0, table, xo, 16u, iters);
    CHK(hipEventRecord(e1, 0));
    CHK(hipDeviceSynchronize());
    float xms;
    CHK(hipEventElapsedTime(&xms, e0, e1));
    // chip-wide rates; per addition: time of one wave-addition on one of the 4 wave slots of a SIMD = 4096 waves x 64 lanes / rate
    const double x_rate = (double)nt * 16 * iters / (xms * 1e-3);
    const double x_ns = 1e9 / x_rate * nt;  // ns per addition and lane at 4 waves per SIMD
    printf("xyzz mixed addition (8M + 2S):             %6.2f G additions/s  (%7.1f ns per addition and lane with 4 waves per SIMD: %.0f cycles per wave-addition on a SIMD)\n",
           x_rate * 1e-9, x_ns, x_ns * 1e-9 * clk / 4);
    std::vector<G1X> xh((size_t)16 * nt);
    CHK(hipMemcpy(xh.data(), xo, xh.size() * sizeof(G1X), hipMemcpyDeviceToHost));
    // ---- batched affine
    std::vector<G1Affine> a16;
    const double t16 = run_affine<16>(table, waves, iters, &a16), t64 = run_affine<64>(table, waves, iters, nullptr),
                 t256 = run_affine<256>(table, waves, 4, nullptr);
    const double n16 = t16 * 1e6 / (16.0 * iters), n64 = t64 * 1e6 / (64.0 * iters), n256 = t256 * 1e6 / (256.0 * 4);
    printf("batched affine, B =  16 sums per lane:     %6.2f G additions/s  (%7.1f ns per addition and lane, %.2f x xyzz)\n", nt / n16, n16, n16 / x_ns);
    printf("batched affine, B =  64 sums per lane:     %6.2f G additions/s  (%7.1f ns per addition and lane, %.2f x xyzz)\n", nt / n64, n64, n64 / x_ns);
    printf("batched affine, B = 256 sums per lane:     %6.2f G additions/s  (%7.1f ns per addition and lane, %.2f x xyzz)\n", nt / n256, n256, n256 / x_ns);
    // t(B) = a + I / B from B = 16 and B = 64
    const double I = (n16 - n64) / (1.0 / 16 - 1.0 / 64), a = n64 - I / 64;
    printf("fit t(B) = a + I / B:  a = %.1f ns (the floor with a FREE inversion: %.2f x xyzz),  I = %.0f ns per inversion = %.1f xyzz additions\n", a,
           a / x_ns, I, I / x_ns);
    if (a < x_ns)
        printf("break-even batch per lane with this (Fermat) inversion: B = %.0f;  with an inversion of 50 products (~ %.0f ns): B = %.0f\n",
               I / (x_ns - a), 50.0 * x_ns / 9.6, 50.0 * x_ns / 9.6 / (x_ns - a));
    else
        printf("no break-even: the batched affine addition is slower than the xyzz addition even with a free inversion\n");
    // ---- same sums?
    size_t bad = 0;
    for (uint32_t c = 0; c < 16; c++)
        for (uint32_t t = 0; t < nt; t += 997) {
            const G1Affine w = norm_host(xh[(size_t)c * nt + t]);
            if (memcmp(&w, &a16[(size_t)c * nt + t], sizeof(G1Affine)) != 0) bad++;
        }
    printf("check: affine sums == xyzz sums on sampled lanes: %s\n", bad ? "MISMATCH" : "ok");
    return bad ? 1 : 0;
}
