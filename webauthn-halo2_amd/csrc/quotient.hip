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

// One row's share of the y-combination: slice `sl` of `ns` takes the gates, permutation chunks and lookups whose index is
// sl mod ns (every group sum is linear in its terms, so the slices' results simply add up).  Returns
// acc + l_0 s0 + l_last sl + l_active sa of the share: (<= 108 ; normalised).  SLICED = false is the whole row (sl = 0, ns = 1).
// C3: the row index is coset-major over three cosets (poly.hip "three cosets"): i = j n + r is the point zeta w_ext^(4 r + j), a
// rotation moves r inside the coset, and the coset points are read from the 4n table with a stride.
template <bool SLICED, bool C3 = false>
__device__ __forceinline__ Fr29 quotient_row(const QuotientArgs& a, uint32_t i, uint32_t sl, uint32_t ns) {
    const uint32_t N = 1u << a.log_ext;
    const uint32_t mask = N - 1;
    const uint32_t nmask = (N >> 2) - 1, cbase = i & ~nmask;  // (C3) the row's coset
    auto rot = [&](int r) {  // two's complement wraps correctly mod N / mod n
        return C3 ? (cbase | ((i + (uint32_t)r) & nmask)) : ((i + (uint32_t)(r * 4)) & mask);
    };

    const Fr29 beta = to29(a.beta), gamma = to29(a.gamma);  // (1 ; 29): the host passes 32 x the challenge (internal form)
    const Fr29 one = const_pow2_29<261, FrParams>();
    const Fr* __restrict__ yp = a.ypow;
    Fr29 zero;
#pragma unroll
    for (int l = 0; l < 9; l++) zero.l[l] = 0;
    Fr29 acc = zero, s0 = zero, sl_ = zero, sa = zero;  // gate terms; l_0 terms; l_last terms; active-row terms
    uint32_t nacc = 0, n0 = 0, nl = 0, na = 0;
    // e: limbs < 2^31.4, value bound <= 168 -> e * y^(..) < 2p.  `term` = the term's place in halo2's order
    auto add_to = [&](Fr29& sum, uint32_t& cnt, const Fr29& e, uint32_t term) { q_acc(sum, cnt, mul29(e, to29(yp[term]))); };

    // ---- gates: term j
    for (uint32_t j = sl; j < a.n_gate; j += ns) {
        // fx_sel[j] = fixed column | form << 24: the gate's selector after halo2's compress_selectors (pk.h Layout::gate_sel):
        // form 0: q;  1: q (2 - q), the used selector of a combined pair;  2: q (1 - q), the never-enabled one — zero on the
        // domain but not on the coset: the gate over the (blinded, otherwise empty) column contributes to h(X)
        const uint32_t form = a.fx_sel[j] >> 24;
        const Fr* c = a.adv[j];
        const Fr29 a0 = q_load(c + i), a1 = q_load(c + rot(1)), a2 = q_load(c + rot(2)), a3 = q_load(c + rot(3));
        Fr29 q = q_load(a.fix[a.fx_sel[j] & 0xffffffu] + i);
        if (form) q = mul29(q, sub29<33, 29>(form == 1 ? add29(one, one) : one, q));  // 32 * 35 = 1120: (8 ; 29)
        const Fr29 m = mul29(a1, a2);                        // 32 * 32 = 1024: (8 ; 29)
        const Fr29 w = sub29<33, 29>(add29(a0, m), a3);      // (40 ; 30) - (32 ; 29): (73 ; 31.3)
        const Fr29 g = mul29(q, w);                          // 32 * 73 = 2336: (15 ; 29)
        add_to(acc, nacc, g, j);
    }

    // ---- permutation: terms n_gate (l0 (1 - z0)), + 1 (l_last ..), + 2 .. (chunk links), then one per chunk
    {
        const uint32_t t_perm = a.n_gate;
        const uint32_t t_link = t_perm + 2, t_prod = t_link + (a.n_chunks - 1);
        if (sl == 0) {
            const Fr29 z0 = q_load(a.z[0] + i);
            add_to(s0, n0, sub29<33, 29>(one, z0), t_perm);              // (34 ; 31)
            const Fr29 zl = q_load(a.z[a.n_chunks - 1] + i);
            add_to(sl_, nl, sub29<33, 29>(sqr29(zl), zl), t_perm + 1);   // (8 ; 29) - (32 ; 29): (41 ; 31)
        }
        for (uint32_t c = SLICED ? (sl ? sl : ns) : 1; c < a.n_chunks; c += ns) {
            const Fr29 zc = q_load(a.z[c] + i);
            const Fr29 zp = q_load(a.z[c - 1] + rot(a.last_rot));
            add_to(s0, n0, sub29<33, 29>(zc, zp), t_link + c - 1);       // (65 ; 31)
        }
        // beta * x with x = zeta * w_ext^i (resident vector), then times delta per column
        const Fr29 delta = to29(a.delta);
        const Fr29 bx0 = mul29(q_load(a.xs + (C3 ? 4 * (i & nmask) + (i >> (a.log_ext - 2)) : i)), beta);      // (2 ; 29)
        Fr29 bx = bx0;
        for (uint32_t c = sl; c < a.n_chunks; c += ns) {
            Fr29 left = q_load(a.z[c] + rot(1));             // (32 ; 29), then <= (8 ; 29)
            Fr29 right = q_load(a.z[c] + i);
            const uint32_t lo = c * a.chunk_len;
            const uint32_t hi = min(a.n_perm, lo + a.chunk_len);
            if (SLICED) bx = mul29(bx0, to29(a.delta_chunk[c]));  // beta x delta^lo: (2 ; 29)
            for (uint32_t p = lo; p < hi; p++) {
                const Fr29 vg = add29(q_load(a.perm_val[p] + i), gamma);              // (33 ; 30)
                const Fr29 bs = mul29(beta, q_load(a.sigma[p] + i));                  // (2 ; 29)
                left = mul29(left, add29(vg, bs));           // 32 * 35 = 1120: (8 ; 29); then 8 * 35: (3 ; 29)
                right = mul29(right, add29(vg, bx));
                if (p + 1 < a.n_perm) bx = mul29(bx, delta);  // (2 ; 29)
            }
            add_to(sa, na, sub29<9, 29>(left, right), t_prod + c);       // (17 ; 31)
        }
    }

    // ---- lookups: five terms each
    const uint32_t t_lk = a.n_gate + 2 + (a.n_chunks - 1) + a.n_chunks;
    for (uint32_t l = sl; l < a.n_lookups; l += ns) {
        const uint32_t t = t_lk + 5 * l;
        const Fr29 z = q_load(a.lk_z[l] + i), zn = q_load(a.lk_z[l] + rot(1));
        const Fr29 pa = q_load(a.lk_a[l] + i), pam = q_load(a.lk_a[l] + rot(-1));
        const Fr29 ps = q_load(a.lk_s[l] + i);
        Fr29 inp;
        if (a.single) inp = mul29(q_load(a.fix[a.fx_qlookup] + i), q_load(a.adv[0] + i));  // (8 ; 29)
        else inp = q_load(a.lk_in[l] + i);                                                   // (32 ; 29)
        const Fr29 tab = q_load(a.fix[a.fx_table] + i);
        add_to(s0, n0, sub29<33, 29>(one, z), t);
        add_to(sl_, nl, sub29<33, 29>(sqr29(z), z), t + 1);
        const Fr29 left = mul29(mul29(zn, add29(pa, beta)), add29(ps, gamma));     // 32 * 33: (8 ; 29); 8 * 33: (3 ; 29)
        const Fr29 right = mul29(mul29(z, add29(inp, beta)), add29(tab, gamma));
        add_to(sa, na, sub29<4, 29>(left, right), t + 2);    // (7 ; 31)
        const Fr29 d = sub29<33, 29>(pa, ps);                // (65 ; 31.3)
        add_to(s0, n0, d, t + 3);
        const Fr29 dd = mul29(norm29(d), sub29<33, 29>(pa, pam));  // 65 * 65 = 4225: (26 ; 29)
        add_to(sa, na, dd, t + 4);
    }

    // sums are (<= 66 ; normalised): each times its multiplier (32) is < 14 p
    const Fr29 t0 = mul29(s0, q_load(a.l0 + i));
    const Fr29 t1 = mul29(sl_, q_load(a.l_last + i));
    const Fr29 t2 = mul29(sa, q_load(a.l_active + i));
    return norm29(add29(add29(acc, t0), add29(t1, t2)));  // 66 + 3 * 14 = 108 <= 168
}

template <bool C3>
__global__ __launch_bounds__(256) void quotient_kernel(const QuotientArgs* __restrict__ ap) {
    const QuotientArgs& a = *ap;
    const uint32_t N = 1u << a.log_ext;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= (C3 ? 3 * (N >> 2) : N)) return;
    const Fr29 total = quotient_row<false, C3>(a, i, 0, 1);
    // times 1/(X^n - 1) (or 1) in the standard form: the product is the standard form of the result, < 2p
    Fr r = from29(mul29(total, to29(a.t_inv[C3 ? i >> (a.log_ext - 2) : i & 3])));
    reduce_once(r);
    fe_store(a.out + i, r);
}

// The many-column shapes (k <= 14: hundreds of gates and lookups over a few thousand rows) have too few rows to fill the chip
// with one lane per row: 2^log_ns lanes share a row, each takes every 2^log_ns-th gate / chunk / lookup, and the shares are
// added through LDS (each first brought below 2p by a product with "one").  Lanes of a workgroup: row-major within a slice.
template <bool C3>
__global__ __launch_bounds__(256) void quotient_sliced_kernel(const QuotientArgs* __restrict__ ap, uint32_t log_ns) {
    __shared__ uint32_t part[256 * 9];
    const QuotientArgs& a = *ap;
    const uint32_t ns = 1u << log_ns, rows = 256u >> log_ns;
    const uint32_t row = threadIdx.x & (rows - 1), sl = threadIdx.x >> (8 - log_ns);
    const uint32_t i = blockIdx.x * rows + row;  // N (and 3 N / 4) is a multiple of 256: no partial workgroups
    const Fr29 share = mul29(quotient_row<true, C3>(a, i, sl, ns), const_pow2_29<261, FrParams>());  // (2 ; 29)
#pragma unroll
    for (int l = 0; l < 9; l++) part[l * 256 + threadIdx.x] = share.l[l];
    __syncthreads();
    if (sl) return;
    Fr29 total = share;
    for (uint32_t s = 1; s < ns; s++) {
        Fr29 v;
#pragma unroll
        for (int l = 0; l < 9; l++) v.l[l] = part[l * 256 + s * rows + row];
        total = norm29(add29(total, v));  // <= 2 * 16 p
    }
    Fr r = from29(mul29(total, to29(a.t_inv[C3 ? i >> (a.log_ext - 2) : i & 3])));
    reduce_once(r);
    fe_store(a.out + i, r);
}

// `d_args` is the argument block in device memory (too large for a kernarg segment)
// `log_slices`: lanes per row (quotient_log_slices); 0 = one lane per row
// `cosets3`: the rows are the [3][n] coset-major rows of the three-coset route (every operand in that layout; 3 n rows)
void launch_quotient_dev(const QuotientArgs* d_args, uint32_t log_ext, uint32_t log_slices, hipStream_t st, bool cosets3) {
    const uint32_t N = cosets3 ? 3u << (log_ext - 2) : 1u << log_ext;
    if (log_slices == 0 || N < 256 || (N & 255)) {
        if (cosets3) hipLaunchKernelGGL(quotient_kernel<true>, dim3((N + 255) / 256), dim3(256), 0, st, d_args);
        else hipLaunchKernelGGL(quotient_kernel<false>, dim3((N + 255) / 256), dim3(256), 0, st, d_args);
    } else {
        if (cosets3) hipLaunchKernelGGL(quotient_sliced_kernel<true>, dim3(N >> (8 - log_slices)), dim3(256), 0, st, d_args, log_slices);
        else hipLaunchKernelGGL(quotient_sliced_kernel<false>, dim3(N >> (8 - log_slices)), dim3(256), 0, st, d_args, log_slices);
    }
}

// lanes per row: enough for ~2^16 lanes in all (measured with whole proofs, tools/quot_ab.sh: k = 11 15.2 -> 13.2 ms, k = 12 10.1 -> 9.2, k = 13 7.8 -> 7.3), at most 16, and not more than there are gates to share out
uint32_t quotient_log_slices(uint32_t log_ext, uint32_t n_gate) {
    uint32_t ls = 0;
#ifndef ZK_QUOT_LANES_LOG
#define ZK_QUOT_LANES_LOG 16
#endif
    while (ls < 4 && log_ext + ls < ZK_QUOT_LANES_LOG && (2u << ls) <= n_gate) ls++;
    return ls;
}

}  // namespace zk
