// ubench_msm_tail.hip — one fixed-base MSM of 2^19 random scalars through the engine's own msm_run, with per-workgroup
// wall-clock stamps of the second-level gather (ZK_TAIL_TRACE): when each workgroup starts, how long its serial part and
// its tree take.  Coordinates are random field elements (the formulas do not test curve membership).  Build:
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -DZK_TAIL_TRACE -I webauthn-halo2_amd/csrc tools/ubench_msm_tail.hip -o tools/ubench_msm_tail
#include "msm.hip"

#include <algorithm>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
using namespace zk;

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

int main() {
    const uint32_t k = 19, n = 1u << k, c = 13, nwin = msm_num_windows(c);
    std::vector<uint32_t> hs((size_t)n * 8), hb((size_t)n * 16);
    srand(11);
    for (auto& w : hs) w = ((uint32_t)rand() << 16) ^ (uint32_t)rand();
    for (size_t i = 0; i < n; i++) hs[i * 8 + 7] &= 0x0fffffffu;  // < r
    for (auto& w : hb) w = ((uint32_t)rand() << 16) ^ (uint32_t)rand();
    for (size_t i = 0; i < (size_t)n * 2; i++) hb[i * 8 + 7] &= 0x0fffffffu;
    Fr* dscal;
    G1Affine *dbases, *dtable;
    CHK(hipMalloc(&dscal, (size_t)n * 32));
    CHK(hipMalloc(&dbases, (size_t)n * 64));
    CHK(hipMalloc(&dtable, (size_t)n * 64 * nwin));
    CHK(hipMemcpy(dscal, hs.data(), (size_t)n * 32, hipMemcpyHostToDevice));
    CHK(hipMemcpy(dbases, hb.data(), (size_t)n * 64, hipMemcpyHostToDevice));
    hipStream_t st;
    CHK(hipStreamCreate(&st));
    CHK(msm_build_table(dbases, n, c, dtable, st));
    hipError_t err;
    MsmWorkspace* ws = msm_workspace_create(n, c, &err, 1);
    CHK(err);
    G1X* hsum;
    CHK(hipHostMalloc(&hsum, 4096 * sizeof(G1X)));
    const Fr* list[1] = {dscal};
    for (int rep = 0; rep < 3; rep++) {
        uint32_t nw, cc;
        CHK(msm_run(ws, list, 1, dbases, n, st, hsum, &nw, &cc, nullptr, dtable, n));
        CHK(hipStreamSynchronize(st));
        static unsigned long long tr[3][8192];
        CHK(hipMemcpyFromSymbol(tr, HIP_SYMBOL(zk_wg_trace), sizeof(tr)));
        // the gather kernel's active workgroups are the first nb * 16 / 256 (part 0) + a few of the later parts
        unsigned long long t0 = ~0ull, t1 = 0;
        std::vector<double> start, serial, tree;
        for (int b = 0; b < 8192; b++) {
            if (tr[2][b] == 0 || tr[2][b] < tr[0][b]) continue;
            t0 = std::min(t0, tr[0][b]);
            t1 = std::max(t1, tr[2][b]);
        }
        int cnt = 0, slow = -1;
        double worst = 0;
        for (int b = 0; b < 8192; b++) {
            if (tr[2][b] == 0 || tr[2][b] < tr[0][b] || tr[0][b] < t0) continue;
            cnt++;
            start.push_back((tr[0][b] - t0) * 0.01);
            serial.push_back((tr[1][b] - tr[0][b]) * 0.01);
            tree.push_back((tr[2][b] - tr[1][b]) * 0.01);
            if ((tr[2][b] - t0) * 0.01 > worst) worst = (tr[2][b] - t0) * 0.01, slow = b;
        }
        auto pct = [](std::vector<double> v, double p) { std::sort(v.begin(), v.end()); return v[(size_t)(p * (v.size() - 1))]; };
        printf("gather: %d workgroups stamped, span %.1f us; start offset p50 %.1f p90 %.1f max %.1f; serial p50 %.1f p90 %.1f max %.1f; tree p50 %.1f p90 %.1f max %.1f; last to finish: workgroup %d (start %.1f serial %.1f tree %.1f)\n",
               cnt, (t1 - t0) * 0.01, pct(start, 0.5), pct(start, 0.9), pct(start, 1.0), pct(serial, 0.5), pct(serial, 0.9), pct(serial, 1.0), pct(tree, 0.5),
               pct(tree, 0.9), pct(tree, 1.0), slow, slow >= 0 ? (tr[0][slow] - t0) * 0.01 : 0, slow >= 0 ? (tr[1][slow] - tr[0][slow]) * 0.01 : 0,
               slow >= 0 ? (tr[2][slow] - tr[1][slow]) * 0.01 : 0);
    }
    return 0;
}
