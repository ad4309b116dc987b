// poly.hip — streaming polynomial kernels over BN254 Fr and the SRS generator.
//
// Device replacements for halo2_proofs `arithmetic::eval_polynomial` and the
// secret-known `ParamsKZG::setup` (reference call
// sites halo2-circuits/src/ecc/ecdsa_p256.rs:258,338,388 via gen_srs, and the
// evaluation phase of create_proof at :366-373,416-423).
#include "engine.h"
#include "prover.h"

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
__global__ __launch_bounds__(256) void poly_eval_batch_kernel(const EvalItem* __restrict__ items, uint32_t n,
                                                              Fr* __restrict__ block_out) {
    __shared__ Fr sh[256];
    const uint32_t e = blockIdx.y;
    const Fr* __restrict__ c = items[e].poly;
    const Fr x = items[e].x, y = items[e].y;
    const uint32_t S = gridDim.x * 256;
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    Fr acc = Fr::zero();
    if (t < n) {
        uint32_t m = (n - 1 - t) / S;
        acc = fe_load(c + t + m * S);
        while (m-- > 0) acc = fe_add(fe_mul(acc, y), fe_load(c + t + m * S));
        Fr xp = Fr::one();
        Fr base = x;
        for (uint32_t k = t; k; k >>= 1) {
            if (k & 1) xp = fe_mul(xp, base);
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
    if (threadIdx.x == 0) fe_store(block_out + e * gridDim.x + blockIdx.x, sh[0]);
}

__global__ __launch_bounds__(256) void poly_sum_batch_kernel(const Fr* __restrict__ in, uint32_t m, Fr* __restrict__ out) {
    __shared__ Fr sh[256];
    const Fr* src = in + (size_t)blockIdx.x * m;
    Fr acc = Fr::zero();
    for (uint32_t i = threadIdx.x; i < m; i += 256) acc = fe_add(acc, fe_load(src + i));
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) sh[threadIdx.x] = fe_add(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) fe_store(out + blockIdx.x, sh[0]);
}

// host fills h_items[e].{poly,x}; y = x^S is derived here.  scratch: count * blocks elements; out: count.
void launch_eval_batch(EvalItem* h_items, EvalItem* d_items, uint32_t count, uint32_t n, Fr* scratch, Fr* out,
                       hipStream_t st) {
    const uint32_t blocks = eval_blocks(n);
    const uint32_t S = blocks * 256;
    for (uint32_t e = 0; e < count; e++) {
        Fr y = Fr::one(), base = h_items[e].x;
        for (uint32_t k = S; k; k >>= 1) {
            if (k & 1) y = fe_mul(y, base);
            base = fe_sqr(base);
        }
        h_items[e].y = y;
    }
    hipMemcpyAsync(d_items, h_items, (size_t)count * sizeof(EvalItem), hipMemcpyHostToDevice, st);
    hipLaunchKernelGGL(poly_eval_batch_kernel, dim3(blocks, count), dim3(256), 0, st, d_items, n, scratch);
    hipLaunchKernelGGL(poly_sum_batch_kernel, dim3(count), dim3(256), 0, st, scratch, blocks, out);
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
