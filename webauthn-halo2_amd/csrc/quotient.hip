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
// 31 field products at k=19 — the kernel is bound by the integer multiplier, not by HBM.
#include "prover.h"
#include "field29.hip.h"

namespace zk {

typedef Fe29<FrParams> Fr29;

// The y-combination is sum_j term_j y^(T-1-j) (halo2 folds it as a Horner chain, one product by y per term).  Terms that
// share a multiplier are grouped here — all l_0 terms, all l_last terms, all active-row terms — so that a term costs one
// product by its (host-computed) power of y instead of two (multiplier + Horner step), and the multiplier is applied once
// per group: 31 field products per row at k=19 instead of 36.  ypow[j] = y^(T-1-j) in term order.
//
// Round 2: the whole row is computed on the carry-free 9 x 29-bit field (field29.hip.h), lazily.  A coset value v (standard
// memory form, canonical) is read as the limbs of 32 v — its internal form x * 2^261 with the value bound k = 32 — for free;
// the constants of the argument block (beta, gamma, delta, the powers of y) arrive already multiplied by 32 (canonical
// internal form, k = 1); the one product that leaves the row (by 1/(X^n - 1), or by 1) takes its constant in the STANDARD form,
// which lands the result in the standard form.  227 instead of 318 instructions per product, 9 per addition instead of 33.
// Bounds are written as (value < k p ; limb bits); a product needs limb products < 2^60.6 and gives limbs < 2^29 and
// value < p (1 + k_a k_b / 169.4); sub29<K, 29>(a, b) needs b normalised and b < (K - 1) p.
__device__ __forceinline__ Fr29 q_load(const Fr* p) { return to29_x32(fe_load(p)); }  // (32 ; 29)

// sum += term with term < 2p: the sum is kept normalised and is folded back below 2p every 32 terms (one product by "one")
__device__ __forceinline__ void q_acc(Fr29& sum, uint32_t& cnt, const Fr29& term) {
    sum = norm29(add29(sum, term));
    if (++cnt == 32) {  // uniform across the launch: the same terms in every row
        sum = mul29(sum, const_pow2_29<261, FrParams>());  // (66 ; 29) -> (2 ; 29)
        cnt = 1;
    }
}

__global__ __launch_bounds__(256) void quotient_kernel(const QuotientArgs* __restrict__ ap) {
    const QuotientArgs& a = *ap;
    const uint32_t N = 1u << a.log_ext;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const uint32_t mask = N - 1;
    auto rot = [&](int r) { return (i + (uint32_t)(r * 4)) & mask; };  // two's complement wraps correctly mod N

    const Fr29 beta = to29(a.beta), gamma = to29(a.gamma);  // (1 ; 29): the host passes 32 x the challenge (internal form)
    const Fr29 one = const_pow2_29<261, FrParams>();
    const Fr* __restrict__ yp = a.ypow;
    uint32_t term = 0;
    Fr29 zero;
#pragma unroll
    for (int l = 0; l < 9; l++) zero.l[l] = 0;
    Fr29 acc = zero, s0 = zero, sl = zero, sa = zero;  // gate terms; l_0 terms; l_last terms; active-row terms
    uint32_t nacc = 0, n0 = 0, nl = 0, na = 0;
    // e: limbs < 2^31.4, value bound <= 168 -> e * y^(..) < 2p
    auto add_to = [&](Fr29& sum, uint32_t& cnt, const Fr29& e) { q_acc(sum, cnt, mul29(e, to29(yp[term++]))); };

    // ---- gates
    for (uint32_t j = 0; j < a.n_gate; j++) {
        if (a.fx_sel[j] == NO_SELECTOR) {  // never-enabled gate: contributes 0 but keeps its power of y
            term++;
            continue;
        }
        const Fr* c = a.adv[j];
        const Fr29 a0 = q_load(c + i), a1 = q_load(c + rot(1)), a2 = q_load(c + rot(2)), a3 = q_load(c + rot(3));
        const Fr29 q = q_load(a.fix[a.fx_sel[j]] + i);
        const Fr29 m = mul29(a1, a2);                        // 32 * 32 = 1024: (8 ; 29)
        const Fr29 w = sub29<33, 29>(add29(a0, m), a3);      // (40 ; 30) - (32 ; 29): (73 ; 31.3)
        const Fr29 g = mul29(q, w);                          // 32 * 73 = 2336: (15 ; 29)
        add_to(acc, nacc, g);
    }

    // ---- permutation
    {
        const Fr29 z0 = q_load(a.z[0] + i);
        add_to(s0, n0, sub29<33, 29>(one, z0));              // (34 ; 31)
        const Fr29 zl = q_load(a.z[a.n_chunks - 1] + i);
        add_to(sl, nl, sub29<33, 29>(sqr29(zl), zl));        // (8 ; 29) - (32 ; 29): (41 ; 31)
        for (uint32_t c = 1; c < a.n_chunks; c++) {
            const Fr29 zc = q_load(a.z[c] + i);
            const Fr29 zp = q_load(a.z[c - 1] + rot(a.last_rot));
            add_to(s0, n0, sub29<33, 29>(zc, zp));           // (65 ; 31)
        }
        // beta * x with x = zeta * w_ext^i (resident vector), then times delta per column
        const Fr29 delta = to29(a.delta);
        Fr29 bx = mul29(q_load(a.xs + i), beta);             // (2 ; 29)
        for (uint32_t c = 0; c < a.n_chunks; c++) {
            Fr29 left = q_load(a.z[c] + rot(1));             // (32 ; 29), then <= (8 ; 29)
            Fr29 right = q_load(a.z[c] + i);
            const uint32_t lo = c * a.chunk_len;
            const uint32_t hi = min(a.n_perm, lo + a.chunk_len);
            for (uint32_t p = lo; p < hi; p++) {
                const Fr29 vg = add29(q_load(a.perm_val[p] + i), gamma);              // (33 ; 30)
                const Fr29 bs = mul29(beta, q_load(a.sigma[p] + i));                  // (2 ; 29)
                left = mul29(left, add29(vg, bs));           // 32 * 35 = 1120: (8 ; 29); then 8 * 35: (3 ; 29)
                right = mul29(right, add29(vg, bx));
                if (p + 1 < a.n_perm) bx = mul29(bx, delta);  // (2 ; 29)
            }
            add_to(sa, na, sub29<9, 29>(left, right));       // (17 ; 31)
        }
    }

    // ---- lookups
    for (uint32_t l = 0; l < a.n_lookups; l++) {
        const Fr29 z = q_load(a.lk_z[l] + i), zn = q_load(a.lk_z[l] + rot(1));
        const Fr29 pa = q_load(a.lk_a[l] + i), pam = q_load(a.lk_a[l] + rot(-1));
        const Fr29 ps = q_load(a.lk_s[l] + i);
        Fr29 inp;
        if (a.single) inp = mul29(q_load(a.fix[a.fx_qlookup] + i), q_load(a.adv[0] + i));  // (8 ; 29)
        else inp = q_load(a.lk_in[l] + i);                                                   // (32 ; 29)
        const Fr29 tab = q_load(a.fix[a.fx_table] + i);
        add_to(s0, n0, sub29<33, 29>(one, z));
        add_to(sl, nl, sub29<33, 29>(sqr29(z), z));
        const Fr29 left = mul29(mul29(zn, add29(pa, beta)), add29(ps, gamma));     // 32 * 33: (8 ; 29); 8 * 33: (3 ; 29)
        const Fr29 right = mul29(mul29(z, add29(inp, beta)), add29(tab, gamma));
        add_to(sa, na, sub29<4, 29>(left, right));           // (7 ; 31)
        const Fr29 d = sub29<33, 29>(pa, ps);                // (65 ; 31.3)
        add_to(s0, n0, d);
        const Fr29 dd = mul29(norm29(d), sub29<33, 29>(pa, pam));  // 65 * 65 = 4225: (26 ; 29)
        add_to(sa, na, dd);
    }

    // sums are (<= 66 ; normalised): each times its multiplier (32) is < 14 p
    const Fr29 t0 = mul29(s0, q_load(a.l0 + i));
    const Fr29 t1 = mul29(sl, q_load(a.l_last + i));
    const Fr29 t2 = mul29(sa, q_load(a.l_active + i));
    const Fr29 total = norm29(add29(add29(acc, t0), add29(t1, t2)));  // 66 + 3 * 14 = 108 <= 168
    // times 1/(X^n - 1) (or 1) in the standard form: the product is the standard form of the result, < 2p
    Fr r = from29(mul29(total, to29(a.t_inv[i & 3])));
    reduce_once(r);
    fe_store(a.out + i, r);
}
// `d_args` is the argument block in device memory (too large for a kernarg segment)
void launch_quotient_dev(const QuotientArgs* d_args, uint32_t log_ext, hipStream_t st) {
    const uint32_t N = 1u << log_ext;
    hipLaunchKernelGGL(quotient_kernel, dim3((N + 255) / 256), dim3(256), 0, st, d_args);
}

}  // namespace zk
