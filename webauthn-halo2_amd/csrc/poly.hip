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

// ------------------------------------------------- three cosets (round 6) ----
// A circuit whose quotient has THREE pieces (degree 4: every bench_ecdsa.config row with two or more advice columns, the proving
// server's k = 17 among them) has deg h < 3n, so h is determined by its values on three of the four cosets
// g_j H, g_j = zeta w_4n^j, that make up halo2's extended domain (EvaluationDomain::extended_k = k + 2: a power of two, one coset
// more than the degree needs).  The prover's private path (prover.hip, Prover::transforms / quotient) therefore works on
// [3][n] "coset-major" vectors — v[j n + i] = f(g_j w_n^i), j < 3 — made by three n-point transforms of the coefficients twisted by
// g_j^m, evaluates the quotient numerator on those 3n rows, and recovers the pieces from three n-point inverse transforms:
// with c_j = g_j^n = z i^j (z = zeta^n, i = w_4n^n) the interpolant of coset j is r_j = h0 + c_j h1 + c_j^2 h2 coefficient by
// coefficient (X^n = c_j on the coset), a 3 x 3 system with constant coefficients.  The coefficients of h are unique, so the
// pieces — and every proof byte — equal those of the 4n-point route; a quarter of the transform and quotient work is gone.
// (The public zk_coeff_to_extended / zk_quotient / zk_extended_to_coeff and the key's file image keep halo2's 4n layout.)

// tab[m] = zeta^(m mod 3) * w_4n^(j m) * 2^10 (Montgomery image: as plain words the factor carries 2^266, ntt.hip `pre`)
__global__ __launch_bounds__(256) void coset3_pre_kernel(const Fr* __restrict__ tw_ext, uint32_t n, uint32_t j, Fr z0, Fr z1, Fr z2,
                                                         Fr* __restrict__ tab) {
    const uint32_t m = blockIdx.x * 256 + threadIdx.x;
    if (m >= n) return;
    const uint32_t r = m % 3;
    fe_store(tab + m, fe_mul(fe_load(tw_ext + (size_t)j * m), r == 0 ? z0 : r == 1 ? z1 : z2));  // j m < 4n
}
void launch_coset3_pre(const Fr* tw_ext, uint32_t n, uint32_t j, const Fr zp1024[3], Fr* tab, hipStream_t st) {
    hipLaunchKernelGGL(coset3_pre_kernel, dim3((n + 255) / 256), dim3(256), 0, st, tw_ext, n, j, zp1024[0], zp1024[1], zp1024[2], tab);
}

// dst[j n + i] = src[4 i + j], j < 3: the key's extended cosets (halo2's order) in coset-major order; blockIdx.y = vector
struct Coset3Vecs {
    const Fr* src[32];
    Fr* dst[32];
};
__global__ __launch_bounds__(256) void coset3_relayout_kernel(Coset3Vecs v, uint32_t n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const Fr* __restrict__ s = v.src[blockIdx.y];
    Fr* __restrict__ d = v.dst[blockIdx.y];
#pragma unroll
    for (uint32_t j = 0; j < 3; j++) fe_store(d + (size_t)j * n + i, fe_load(s + 4 * (size_t)i + j));
}
void launch_coset3_relayout(const Fr* const* src, Fr* const* dst, uint32_t count, uint32_t n, hipStream_t st) {
    for (uint32_t c0 = 0; c0 < count; c0 += 32) {
        Coset3Vecs v;
        const uint32_t cnt = count - c0 < 32 ? count - c0 : 32;
        for (uint32_t q = 0; q < cnt; q++) {
            v.src[q] = src[c0 + q];
            v.dst[q] = dst[c0 + q];
        }
        hipLaunchKernelGGL(coset3_relayout_kernel, dim3((n + 255) / 256, cnt), dim3(256), 0, st, v, n);
    }
}

// h[j n + m] = a_j[m] = g_j^m r_j[m] (the three inverse transforms' outputs)  ->  h[k n + m] = h_k[m], k < 3, in place:
//   v_j = a_j[m] w_4n^(-j m) = zeta^m r_j[m];   h1 = (v0 - v2) / (2 z);   A = (v0 + v2) / 2;   B = v1 - z i h1;
//   h0 = (A + B) / 2;   h2 = (A - B) / (2 z^2);   each times zeta^(-m)
__global__ __launch_bounds__(256) void coset3_combine_kernel(Fr* __restrict__ h, const Fr* __restrict__ tw_ext, uint32_t n, Coset3Consts k) {
    const uint32_t m = blockIdx.x * 256 + threadIdx.x;
    if (m >= n) return;
    const uint32_t N4 = 4 * n;
    const Fr v0 = fe_load(h + m);
    const Fr v1 = fe_mul(fe_load(h + (size_t)n + m), fe_load(tw_ext + ((N4 - m) & (N4 - 1))));
    const Fr v2 = fe_mul(fe_load(h + 2 * (size_t)n + m), fe_load(tw_ext + ((N4 - 2 * m) & (N4 - 1))));
    const Fr h1 = fe_mul(fe_sub(v0, v2), k.inv_2z);
    const Fr A = fe_mul(fe_add(v0, v2), k.inv2);
    const Fr B = fe_sub(v1, fe_mul(h1, k.zi));
    const Fr h0 = fe_mul(fe_add(A, B), k.inv2);
    const Fr h2 = fe_mul(fe_sub(A, B), k.inv_2z2);
    const uint32_t r = m % 3;
    const Fr s = r == 0 ? k.zinv[0] : r == 1 ? k.zinv[1] : k.zinv[2];
    fe_store(h + m, fe_mul(h0, s));
    fe_store(h + (size_t)n + m, fe_mul(h1, s));
    fe_store(h + 2 * (size_t)n + m, fe_mul(h2, s));
}
void launch_coset3_combine(Fr* h, const Fr* tw_ext, uint32_t n, const Coset3Consts& k, hipStream_t st) {
    hipLaunchKernelGGL(coset3_combine_kernel, dim3((n + 255) / 256), dim3(256), 0, st, h, tw_ext, n, k);
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
