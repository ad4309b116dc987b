// poly.hip — streaming polynomial kernels over BN254 Fr and the SRS generator.
//
// Device replacements for halo2_proofs `arithmetic::eval_polynomial` and the
// secret-known `ParamsKZG::setup` (reference call
// sites halo2-circuits/src/ecc/ecdsa_p256.rs:258,338,388 via gen_srs, and the
// evaluation phase of create_proof at :366-373,416-423).
#include "engine.h"
#include "prover.h"
#include <string.h>

namespace zk {

// ---------------------------------------------------------------- eval -----
// p(x) = sum_t x^t * ( sum_m c[t + m*S] * (x^S)^m ),  S = total threads.
// Reads are coalesced (consecutive t), one Horner chain per thread in y = x^S.
__global__ __launch_bounds__(256) void poly_eval_partial_kernel(const Fr* __restrict__ c, uint32_t n, Fr x, Fr y,
                                                                Fr* __restrict__ block_out) {
    __shared__ Fr sh[256];
    const uint32_t S = gridDim.x * 256;
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    Fr acc = Fr::zero();
    if (t < n) {
        // highest m with t + m*S < n
        uint32_t m = (n - 1 - t) / S;
        acc = fe_load(c + t + m * S);
        while (m-- > 0) acc = fe_add(fe_mul(acc, y), fe_load(c + t + m * S));
        // times x^t
        Fr xp = Fr::one();
        Fr base = x;
        for (uint32_t e = t; e; e >>= 1) {
            if (e & 1) xp = fe_mul(xp, base);
            base = fe_sqr(base);
        }
        acc = fe_mul(acc, xp);
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) sh[threadIdx.x] = fe_add(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) fe_store(block_out + blockIdx.x, sh[0]);
}

__global__ __launch_bounds__(256) void poly_sum_kernel(const Fr* __restrict__ in, uint32_t m, Fr* __restrict__ out) {
    __shared__ Fr sh[256];
    Fr acc = Fr::zero();
    for (uint32_t i = threadIdx.x; i < m; i += 256) acc = fe_add(acc, fe_load(in + i));
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) sh[threadIdx.x] = fe_add(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) fe_store(out, sh[0]);
}

// scratch: at least eval_blocks(n) + 1 elements; result in scratch[eval_blocks(n)]
uint32_t eval_blocks(uint32_t n) {
    uint32_t b = (n + 256 * 16 - 1) / (256 * 16);
    if (b < 1) b = 1;
    if (b > 1024) b = 1024;
    return b;
}

void launch_eval(const Fr* c, uint32_t n, const Fr& x, Fr* scratch, hipStream_t st) {
    const uint32_t blocks = eval_blocks(n);
    const uint32_t S = blocks * 256;
    Fr y = Fr::one(), base = x;
    for (uint32_t e = S; e; e >>= 1) {
        if (e & 1) y = fe_mul(y, base);
        base = fe_sqr(base);
    }
    hipLaunchKernelGGL(poly_eval_partial_kernel, dim3(blocks), dim3(256), 0, st, c, n, x, y, scratch);
    hipLaunchKernelGGL(poly_sum_kernel, dim3(1), dim3(256), 0, st, scratch, blocks, scratch + blocks);
}

// Batched form: blockIdx.y selects the (polynomial, point) pair; one launch evaluates every
// opened value of a proof (18 at k=19, 43 at k=17, about 3300 at k=11).
// A workgroup owns 256 * M consecutive coefficients: lane t runs a Horner chain in y = x^256 over c[base + t + 256 m]
// (coalesced), and sum_t P_t x^t is folded in LDS from the top: P_t += x^s P_(t+s) for s = 128, 64, .. 1 — eight products
// on shrinking lane sets.  (Round 1 raised x^t per lane by square-and-multiply: 28 of a lane's 44 products.)
__global__ __launch_bounds__(256) void poly_eval_batch_kernel(const EvalItem* __restrict__ items, uint32_t n, uint32_t M,
                                                              Fr* __restrict__ block_out) {
    __shared__ Fr sh[256];
    const uint32_t e = blockIdx.y;
    const Fr* __restrict__ c = items[e].poly;
    const Fr y = items[e].y;
    const uint32_t base = blockIdx.x * 256 * M + threadIdx.x;
    Fr acc = Fr::zero();
#pragma unroll 1
    for (uint32_t m = M; m-- > 0;) {
        const uint32_t idx = base + 256 * m;
        if (idx < n) {  // a lane's indices grow with m: once inside, it stays inside
            const Fr v = fe_load(c + idx);
            acc = fe_add(fe_mul(acc, y), v);
        }
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    const uint32_t live = min(256u, n - min(n, blockIdx.x * 256 * M));  // lanes holding coefficients (a short last workgroup)
#pragma unroll 1
    for (int l = 7; l >= 0; l--) {
        const uint32_t s = 1u << l;
        if (s >= live) continue;  // the upper half is all zero (workgroup-uniform)
        if (threadIdx.x < s) sh[threadIdx.x] = fe_add(sh[threadIdx.x], fe_mul(sh[threadIdx.x + s], items[e].pw[l]));
        __syncthreads();
    }
    if (threadIdx.x == 0) fe_store(block_out + e * gridDim.x + blockIdx.x, sh[0]);
}

// sum_b B_b z^b over the m <= 1024 workgroup results of one evaluation: Horner in z^256 per lane, then the same fold
__global__ __launch_bounds__(256) void poly_sum_batch_kernel(const EvalItem* __restrict__ items, const Fr* __restrict__ in, uint32_t m,
                                                             Fr* __restrict__ out) {
    __shared__ Fr sh[256];
    const uint32_t e = blockIdx.x;
    const Fr* src = in + (size_t)e * m;
    Fr acc = Fr::zero();
#pragma unroll 1
    for (uint32_t q = (m + 255) / 256; q-- > 0;) {
        const uint32_t idx = threadIdx.x + 256 * q;
        if (idx < m) acc = fe_add(fe_mul(acc, items[e].zpw[8]), fe_load(src + idx));
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
#pragma unroll 1
    for (int l = 7; l >= 0; l--) {
        const uint32_t s = 1u << l;
        if (s >= m) continue;  // nothing above lane m - 1
        if (threadIdx.x < s) sh[threadIdx.x] = fe_add(sh[threadIdx.x], fe_mul(sh[threadIdx.x + s], items[e].zpw[l]));
        __syncthreads();
    }
    if (threadIdx.x == 0) fe_store(out + e, sh[0]);
}

// host fills h_items[e].{poly,x}; the powers are derived here.  scratch: count * blocks elements; out: count.
void launch_eval_batch(EvalItem* h_items, EvalItem* d_items, uint32_t count, uint32_t n, Fr* scratch, Fr* out,
                       hipStream_t st) {
    const uint32_t blocks = eval_blocks(n);
    const uint32_t M = (n + blocks * 256 - 1) / (blocks * 256);  // coefficients per lane (16 up to 2^22)
    // a proof opens at a handful of points (x, w x, w^-1 x, ..): the powers are computed once per distinct point
    uint32_t distinct[16], nd = 0;
    for (uint32_t e = 0; e < count; e++) {
        EvalItem& it = h_items[e];
        uint32_t same = count;
        for (uint32_t d = 0; d < nd; d++)
            if (memcmp(&h_items[distinct[d]].x, &it.x, sizeof(Fr)) == 0) {
                same = distinct[d];
                break;
            }
        if (same != count) {
            memcpy(it.pw, h_items[same].pw, sizeof(it.pw) + sizeof(it.y) + sizeof(it.zpw));
            continue;
        }
        if (nd < 16) distinct[nd++] = e;
        it.pw[0] = it.x;
        for (int l = 1; l < 8; l++) it.pw[l] = fe_sqr(it.pw[l - 1]);
        it.y = fe_sqr(it.pw[7]);
        Fr z = Fr::one(), base = it.y;  // z = y^M = x^(256 M)
        for (uint32_t k = M; k; k >>= 1) {
            if (k & 1) z = fe_mul(z, base);
            base = fe_sqr(base);
        }
        it.zpw[0] = z;
        for (int l = 1; l < 9; l++) it.zpw[l] = fe_sqr(it.zpw[l - 1]);
    }
    hipMemcpyAsync(d_items, h_items, (size_t)count * sizeof(EvalItem), hipMemcpyHostToDevice, st);
    hipLaunchKernelGGL(poly_eval_batch_kernel, dim3(blocks, count), dim3(256), 0, st, d_items, n, M, scratch);
    hipLaunchKernelGGL(poly_sum_batch_kernel, dim3(count), dim3(256), 0, st, d_items, scratch, blocks, out);
}

// ------------------------------------------------------------------ SRS ----

// out[i] = L_i(s) = w^i * c / (s - w^i),  c = (s^n - 1)/n   (ParamsKZG::setup, g_lagrange scalars)
__global__ __launch_bounds__(256) void srs_lagrange_scalars_kernel(const Fr* __restrict__ tw, uint32_t n, Fr s, Fr c,
                                                                   Fr* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const Fr wi = fe_load(tw + i);
    const Fr d = fe_sub(s, wi);
    fe_store(out + i, fe_mul(fe_mul(wi, c), fe_inv(d)));
}

// out[i] = [scalars[i]] G1 (affine), G1 = (1, 2); table[w*256 + d] = [d * 256^w] G1 affine
__global__ __launch_bounds__(64) void srs_fixed_base_kernel(const Fr* __restrict__ scalars, uint32_t n,
                                                            const G1Affine* __restrict__ table,
                                                            G1Affine* __restrict__ out) {
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const Fr s = fe_from_mont(fe_load(scalars + i));
    G1X acc = G1X::identity();
#pragma unroll 1
    for (int w = 0; w < 32; w++) {
        const uint32_t byte = (s.v[w >> 2] >> ((w & 3) * 8)) & 0xff;
        if (byte) {
            const G1Affine p = affine_load(table + w * 256 + byte);
            g1x_add_affine(acc, p.x, p.y);
        }
    }
    G1Affine r;
    if (acc.is_identity()) {
        r.x = Fq::zero();
        r.y = Fq::zero();
    } else {
        const Fq t = fe_inv(acc.zzz);           // 1/ZZZ
        const Fq u = fe_mul(acc.zz, t);         // ZZ/ZZZ = 1/Z
        r.x = fe_mul(acc.x, fe_sqr(u));         // X / ZZ
        r.y = fe_mul(acc.y, t);                 // Y / ZZZ
    }
    fe_store(&out[i].x, r.x);
    fe_store(&out[i].y, r.y);
}

void launch_srs_lagrange_scalars(const Fr* tw, uint32_t n, const Fr& s, const Fr& c, Fr* out, hipStream_t st) {
    hipLaunchKernelGGL(srs_lagrange_scalars_kernel, dim3((n + 255) / 256), dim3(256), 0, st, tw, n, s, c, out);
}

void launch_srs_fixed_base(const Fr* scalars, uint32_t n, const G1Affine* table, G1Affine* out, hipStream_t st) {
    hipLaunchKernelGGL(srs_fixed_base_kernel, dim3((n + 63) / 64), dim3(64), 0, st, scalars, n, table, out);
}

}  // namespace zk
