// ntt.hip — radix-2^b Stockham NTT over BN254 Fr for gfx950.
//
// Device replacement for halo2_proofs `arithmetic::best_fft` and the
// `EvaluationDomain::{lagrange_to_coeff, coeff_to_extended, extended_to_coeff}`
// wrappers around it (reference call sites
// halo2-circuits/src/ecc/ecdsa_p256.rs:366-373,416-423).  Natural order in,
// natural order out: out[i] = sum_j in[j] * w^(i*j).
//
// One launch = one Stockham pass of radix R = 2^log_r.  A 256-thread workgroup
// owns T consecutive butterflies-groups j, stages the R x T tile in LDS
// (R*T = 512 elements = 16 KiB, so ~8 workgroups share a CU and hide each other's load latency;
// 2048-element tiles measured 24 % slower), applies the inter-pass twiddle on load, runs
// log_r DIF stages entirely in LDS, and writes the tile out in autosort order.
// Global traffic per pass is one read + one write of the vector; coset scaling
// (zeta^i pre-multiply, zero-extension) is fused into the first pass's load and
// 1/N, coset un-scaling and truncation into the last pass's store.
#include "field.hip.h"
#include <stdlib.h>
#include <string.h>

#include "engine.h"

namespace zk {

static constexpr int NTT_TILE_LOG = 9;   // 512 elements x 32 B = 16 KiB LDS: many workgroups per CU hide the load latency
static constexpr int NTT_THREADS = 256;

struct NttPassArgs {
    const Fr* in[NTT_MAX_BATCH];   // one vector per blockIdx.y
    Fr* out[NTT_MAX_BATCH];
    const Fr* tw;        // w_N^i for i in [0, N)
    uint32_t log_n;
    uint32_t log_r;      // this pass's radix
    uint32_t log_ns;     // product of radices of earlier passes
    uint32_t log_t;      // j's per workgroup
    uint32_t inverse;    // use w^-1 (index N - e)
    uint32_t n_in;       // elements >= n_in read as zero (first pass only; else N)
    uint32_t n_out;      // elements >= n_out are not stored (last pass only; else N)
    uint32_t has_pre;    // multiply in[i] by pre[i % 3] on load (first pass)
    uint32_t has_post;   // multiply out[i] by post[i % 3] on store (last pass)
    Fr pre[3];
    Fr post[3];
};

__global__ __launch_bounds__(NTT_THREADS) void ntt_pass_kernel(const NttPassArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Fr* s = reinterpret_cast<Fr*>(smem_raw);
    const Fr* __restrict__ vin = a.in[blockIdx.y];
    Fr* __restrict__ vout = a.out[blockIdx.y];

    const uint32_t N = 1u << a.log_n;
    const uint32_t R = 1u << a.log_r;
    const uint32_t T = 1u << a.log_t;
    const uint32_t tile = R * T;
    const uint32_t j0 = blockIdx.x * T;
    const uint32_t ns_mask = (1u << a.log_ns) - 1;
    const uint32_t col_stride = N >> a.log_r;               // N / R
    const uint32_t tw_shift = a.log_n - a.log_ns - a.log_r;  // N / (Ns * R)
    const uint32_t nmask = N - 1;

    // ---- the R/2 twiddles of the in-tile stages (w_R^j = w_N^(j N/R)) go to LDS once per workgroup
    Fr* wr = s + tile;
    for (uint32_t j = threadIdx.x; j < (R >> 1); j += NTT_THREADS) {
        const uint32_t ex = j << (a.log_n - a.log_r);
        wr[j] = fe_load(a.tw + (a.inverse ? ((N - ex) & nmask) : ex));
    }

    // ---- load: s[r*T + t] = in[j + r*N/R] * pre * w_{Ns*R}^{r*(j mod Ns)}
    for (uint32_t e = threadIdx.x; e < tile; e += NTT_THREADS) {
        const uint32_t t = e & (T - 1), r = e >> a.log_t;
        const uint32_t j = j0 + t;
        const uint32_t idx = j + r * col_stride;
        Fr x;
        if (idx < a.n_in) {
            x = fe_load(vin + idx);
            if (a.has_pre) {
                const uint32_t m = idx % 3;
                if (m) x = fe_mul(x, a.pre[m]);
            }
            const uint32_t ex = (r * (j & ns_mask)) << tw_shift;
            if (ex) {
                const uint32_t ti = a.inverse ? ((N - ex) & nmask) : ex;
                x = fe_mul(x, fe_load(a.tw + ti));
            }
        } else {
            x = Fr::zero();
        }
        s[e] = x;
    }
    __syncthreads();

    // ---- log_r DIF stages over the r dimension (natural in, bit-reversed out)
    const uint32_t nbf = tile >> 1;
    for (int st = (int)a.log_r - 1; st >= 0; st--) {
        const uint32_t half = 1u << st;
        for (uint32_t b = threadIdx.x; b < nbf; b += NTT_THREADS) {
            const uint32_t t = b & (T - 1), p = b >> a.log_t;
            const uint32_t lo = p & (half - 1);
            const uint32_t i = ((p >> st) << (st + 1)) | lo;
            Fr* pu = s + i * T + t;
            Fr* pv = pu + half * T;
            const Fr u = *pu, v = *pv;
            *pu = fe_add(u, v);
            Fr d = fe_sub(u, v);
            if (lo) d = fe_mul(d, wr[lo << (a.log_r - st - 1)]);  // w_R^(lo * R/(2*half))
            *pv = d;
        }
        __syncthreads();
    }

    // ---- store: out[(j / Ns) * Ns * R + (j mod Ns) + r * Ns] = X_r
    const uint32_t brev_shift = 32 - a.log_r;
    for (uint32_t e = threadIdx.x; e < tile; e += NTT_THREADS) {
        uint32_t t, r;
        if (a.log_ns == 0) {  // dst = j*R + r : contiguous over r
            r = e & (R - 1);
            t = e >> a.log_r;
        } else {              // contiguous over j within an Ns block
            t = e & (T - 1);
            r = e >> a.log_t;
        }
        const uint32_t j = j0 + t;
        const uint32_t dst = ((j >> a.log_ns) << (a.log_ns + a.log_r)) + (j & ns_mask) + (r << a.log_ns);
        if (dst < a.n_out) {
            const uint32_t rb = a.log_r ? (__brev(r) >> brev_shift) : 0;
            Fr x = s[rb * T + t];
            if (a.has_post) x = fe_mul(x, a.post[dst % 3]);
            fe_store(vout + dst, x);
        }
    }
}

// Twiddle table: tw[i] = w^i, i in [0, N).  Each thread seeds w^(i0) by
// square-and-multiply, then walks 64 consecutive powers.
__global__ void ntt_twiddle_kernel(Fr* tw, Fr w, uint32_t n) {
    const uint32_t CH = 64;
    const uint32_t i0 = (blockIdx.x * blockDim.x + threadIdx.x) * CH;
    if (i0 >= n) return;
    Fr cur = Fr::one();
    Fr base = w;
    for (uint32_t e = i0; e; e >>= 1) {
        if (e & 1) cur = fe_mul(cur, base);
        base = fe_sqr(base);
    }
    for (uint32_t k = 0; k < CH && i0 + k < n; k++) {
        fe_store(tw + i0 + k, cur);
        cur = fe_mul(cur, w);
    }
}

void launch_twiddles(Fr* tw, const Fr& w, uint32_t n, hipStream_t st) {
    const uint32_t threads = (n + 63) / 64;
    hipLaunchKernelGGL(ntt_twiddle_kernel, dim3((threads + 63) / 64), dim3(64), 0, st, tw, w, n);
}

// Plan the passes of a 2^log_n transform: radices as even as possible, each <= max_log_r.
int ntt_plan(uint32_t log_n, uint32_t max_log_r, uint32_t bits[8]) {
    if (log_n == 0) return 0;
    const uint32_t np = (log_n + max_log_r - 1) / max_log_r;
    uint32_t rem = log_n;
    for (uint32_t p = 0; p < np; p++) {
        bits[p] = (rem + (np - p) - 1) / (np - p);
        rem -= bits[p];
    }
    return (int)np;
}

// Runs all passes.  `a` is the input (left intact unless it is also `b`/`c`);
// ping-pongs between b and c so that the result lands in `dst`.  dst must be
// distinct from src unless a scratch `tmp` (N elements) is supplied.
hipError_t ntt_run(const NttJob& job, hipStream_t st) {
    const uint32_t log_n = job.log_n;
    const uint32_t N = 1u << log_n;
    const uint32_t batch = job.batch ? job.batch : 1;
    if (batch > NTT_MAX_BATCH) return hipErrorInvalidValue;
    const Fr* srcs[NTT_MAX_BATCH];
    Fr* dsts[NTT_MAX_BATCH];
    for (uint32_t b = 0; b < batch; b++) {
        srcs[b] = job.batch ? job.srcs[b] : job.src;
        dsts[b] = job.batch ? job.dsts[b] : job.dst;
    }
    uint32_t bits[8];
    const int np = ntt_plan(log_n, job.max_log_r ? job.max_log_r : 7, bits);
    if (np == 0) {  // N == 1
        for (uint32_t b = 0; b < batch; b++)
            if (dsts[b] != srcs[b]) {
                hipError_t e = hipMemcpyAsync(dsts[b], srcs[b], sizeof(Fr), hipMemcpyDeviceToDevice, st);
                if (e != hipSuccess) return e;
            }
        return hipSuccess;
    }
    // Ping-pong so that the last pass writes dst and no pass runs in place:
    // pass p writes bufs[(np - 1 - p) & 1] with bufs = {dst, tmp}.
    bool inplace = false;
    for (uint32_t b = 0; b < batch; b++) inplace = inplace || srcs[b] == dsts[b];
    if (job.tmp == nullptr && (np > 1 || inplace)) return hipErrorInvalidValue;
    const Fr* cur_in[NTT_MAX_BATCH];
    for (uint32_t b = 0; b < batch; b++) {
        cur_in[b] = srcs[b];
        if (srcs[b] == dsts[b] && (np & 1)) {
            // first pass would read and write dst: stage the input in tmp
            hipError_t e = hipMemcpyAsync(job.tmp + (size_t)b * N, srcs[b], sizeof(Fr) * (size_t)(job.n_in < N ? job.n_in : N),
                                          hipMemcpyDeviceToDevice, st);
            if (e != hipSuccess) return e;
            cur_in[b] = job.tmp + (size_t)b * N;
        }
    }
    int which = (np & 1) ? 0 : 1;  // 0: dst, 1: tmp
    uint32_t log_ns = 0;
    for (int p = 0; p < np; p++) {
        NttPassArgs a;
        memset(&a, 0, sizeof(a));
        for (uint32_t b = 0; b < batch; b++) {
            a.in[b] = cur_in[b];
            a.out[b] = which ? job.tmp + (size_t)b * N : dsts[b];
        }
        a.tw = job.tw;
        a.log_n = log_n;
        a.log_r = bits[p];
        a.log_ns = log_ns;
        uint32_t log_t = NTT_TILE_LOG - a.log_r;
        if (log_t > log_n - a.log_r) log_t = log_n - a.log_r;
        a.log_t = log_t;
        a.inverse = job.inverse;
        a.n_in = (p == 0) ? job.n_in : N;
        a.n_out = (p == np - 1) ? job.n_out : N;
        a.has_pre = (p == 0) ? job.has_pre : 0;
        a.has_post = (p == np - 1) ? job.has_post : 0;
        for (int i = 0; i < 3; i++) {
            a.pre[i] = job.pre[i];
            a.post[i] = job.post[i];
        }
        const uint32_t blocks = N >> (a.log_r + a.log_t);
        const size_t lds = ((size_t)sizeof(Fr) << (a.log_r + a.log_t)) + (sizeof(Fr) << a.log_r) / 2;
        hipLaunchKernelGGL(ntt_pass_kernel, dim3(blocks, batch), dim3(NTT_THREADS), lds, st, a);
        for (uint32_t b = 0; b < batch; b++) cur_in[b] = a.out[b];
        which ^= 1;
        log_ns += bits[p];
    }
    return hipGetLastError();
}

}  // namespace zk
