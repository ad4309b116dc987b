// quotient.hip — fused evaluation of the quotient numerator over the extended coset.
//
// Device replacement for halo2_proofs `plonk::evaluation::Evaluator::evaluate_h`
// followed by `EvaluationDomain::divide_by_vanishing_poly` (SURVEY.md §8a a6;
// reached from halo2-circuits/src/ecc/ecdsa_p256.rs:366-373,416-423).  The
// expressions and their y-Horner order are those the reference's generated
// verifier checks (proving-server/P256Verifier.yul:406-552):
//   gates        q_j (a_j + a_j(wX) a_j(w^2 X) - a_j(w^3 X))
//   permutation  l0 (1 - z_0);  l_last (z_last^2 - z_last);  l0 (z_i - z_{i-1}(w^last X));
//                active (z_i(wX) prod(v + beta sigma + gamma) - z_i prod(v + beta delta^c X + gamma))
//   lookups      l0 (1 - zL);  l_last (zL^2 - zL);
//                active (zL(wX)(a'+beta)(s'+gamma) - zL (in+beta)(t+gamma));
//                l0 (a' - s');  active (a' - s')(a' - a'(w^-1 X))
// One thread per coset row; every operand is a resident extended-coset vector and a
// rotation by r is the index shift r * 2^(ext_k - k).  The result is multiplied by
// 1/(X^n - 1), which has period 4 on the coset.  Streaming: ~20 x 32 B loads and one 32 B store per row,
// 30 Montgomery products at k=19 — the kernel is bound by the integer multiplier, not by HBM.
#include "prover.h"

namespace zk {

// The y-combination is sum_j term_j y^(T-1-j) (halo2 folds it as a Horner chain, one product by y per term).  Terms that
// share a multiplier are grouped here — all l_0 terms, all l_last terms, all active-row terms — so that a term costs one
// product by its (host-computed) power of y instead of two (multiplier + Horner step), and the multiplier is applied once
// per group: 30 field products per row at k=19 instead of 36.  ypow[j] = y^(T-1-j) in term order.
__global__ __launch_bounds__(256) void quotient_kernel(const QuotientArgs* __restrict__ ap) {
    const QuotientArgs& a = *ap;
    const uint32_t N = 1u << a.log_ext;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const uint32_t mask = N - 1;
    auto rot = [&](int r) { return (i + (uint32_t)(r * 4)) & mask; };  // two's complement wraps correctly mod N

    const Fr beta = a.beta, gamma = a.gamma;
    const Fr* __restrict__ yp = a.ypow;
    uint32_t term = 0;
    Fr acc = Fr::zero();   // gates: every term has its own multiplier (the selector)
    Fr s0 = Fr::zero();    // sum of the l_0 terms (x their powers of y)
    Fr sl = Fr::zero();    // l_last terms
    Fr sa = Fr::zero();    // active-row terms
    auto add_to = [&](Fr& sum, const Fr& e) { sum = fe_add(sum, fe_mul(e, fe_load(yp + term++))); };

    // ---- gates
    for (uint32_t j = 0; j < a.n_gate; j++) {
        if (a.fx_sel[j] == NO_SELECTOR) {  // never-enabled gate: contributes 0 but keeps its power of y
            term++;
            continue;
        }
        const Fr* c = a.adv[j];
        const Fr a0 = fe_load(c + i), a1 = fe_load(c + rot(1)), a2 = fe_load(c + rot(2)), a3 = fe_load(c + rot(3));
        const Fr q = fe_load(a.fix[a.fx_sel[j]] + i);
        add_to(acc, fe_mul(q, fe_sub(fe_add(a0, fe_mul(a1, a2)), a3)));
    }

    const Fr one = Fr::one();

    // ---- permutation
    {
        const Fr z0 = fe_load(a.z[0] + i);
        add_to(s0, fe_sub(one, z0));
        const Fr zl = fe_load(a.z[a.n_chunks - 1] + i);
        add_to(sl, fe_sub(fe_sqr(zl), zl));
        for (uint32_t c = 1; c < a.n_chunks; c++) {
            const Fr zc = fe_load(a.z[c] + i);
            const Fr zp = fe_load(a.z[c - 1] + rot(a.last_rot));
            add_to(s0, fe_sub(zc, zp));
        }
        // beta * x with x = zeta * w_ext^i (resident vector), then times delta per column
        Fr bx = fe_mul(fe_load(a.xs + i), beta);
        for (uint32_t c = 0; c < a.n_chunks; c++) {
            Fr left = fe_load(a.z[c] + rot(1));
            Fr right = fe_load(a.z[c] + i);
            const uint32_t lo = c * a.chunk_len;
            const uint32_t hi = min(a.n_perm, lo + a.chunk_len);
            for (uint32_t p = lo; p < hi; p++) {
                const Fr v = fe_load(a.perm_val[p] + i);
                const Fr vg = fe_add(v, gamma);
                left = fe_mul(left, fe_add(vg, fe_mul(beta, fe_load(a.sigma[p] + i))));
                right = fe_mul(right, fe_add(vg, bx));
                if (p + 1 < a.n_perm) bx = fe_mul(bx, a.delta);
            }
            add_to(sa, fe_sub(left, right));
        }
    }

    // ---- lookups
    for (uint32_t l = 0; l < a.n_lookups; l++) {
        const Fr z = fe_load(a.lk_z[l] + i), zn = fe_load(a.lk_z[l] + rot(1));
        const Fr pa = fe_load(a.lk_a[l] + i), pam = fe_load(a.lk_a[l] + rot(-1));
        const Fr ps = fe_load(a.lk_s[l] + i);
        Fr inp;
        if (a.single) inp = fe_mul(fe_load(a.fix[a.fx_qlookup] + i), fe_load(a.adv[0] + i));
        else inp = fe_load(a.lk_in[l] + i);
        const Fr tab = fe_load(a.fix[a.fx_table] + i);
        add_to(s0, fe_sub(one, z));
        add_to(sl, fe_sub(fe_sqr(z), z));
        const Fr left = fe_mul(fe_mul(zn, fe_add(pa, beta)), fe_add(ps, gamma));
        const Fr right = fe_mul(fe_mul(z, fe_add(inp, beta)), fe_add(tab, gamma));
        add_to(sa, fe_sub(left, right));
        const Fr d = fe_sub(pa, ps);
        add_to(s0, d);
        add_to(sa, fe_mul(d, fe_sub(pa, pam)));
    }

    acc = fe_add(acc, fe_mul(s0, fe_load(a.l0 + i)));
    acc = fe_add(acc, fe_mul(sl, fe_load(a.l_last + i)));
    acc = fe_add(acc, fe_mul(sa, fe_load(a.l_active + i)));
    fe_store(a.out + i, a.divide ? fe_mul(acc, a.t_inv[i & 3]) : acc);
}

// `d_args` is the argument block in device memory (too large for a kernarg segment)
void launch_quotient_dev(const QuotientArgs* d_args, uint32_t log_ext, hipStream_t st) {
    const uint32_t N = 1u << log_ext;
    hipLaunchKernelGGL(quotient_kernel, dim3((N + 255) / 256), dim3(256), 0, st, d_args);
}

}  // namespace zk
