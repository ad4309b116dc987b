// vkrepr.h — `VerifyingKey::transcript_repr` as halo2 computes it (host code; used by zk_keygen and zk_pk_read).
//
// halo2_proofs plonk.rs `VerifyingKey::from_parts` [RECALLED, then pinned]:
//     s = format!("{:?}", vk.pinned());
//     transcript_repr = Fr::from_bytes_wide(blake2b-512(personal "Halo2-Verify-Key")(s.len() as u64 LE || s))
// The string is the `Debug` rendering of PinnedVerificationKey { base_modulus, scalar_modulus, domain, cs,
// fixed_commitments, permutation } with `cs` the constraint system after selector compression — here the one
// halo2-lib's RangeConfig / FlexGateConfig build for the reference's ECDSA circuit (halo2-circuits/src/ecc/
// ecdsa_p256.rs:129-139 via FpConfig::configure): lookup table column first (fixed column 0), then the constants
// columns, per gate column a selector and the gate q * (a + b * c - out); compress_selectors turns every selector
// into a fixed-column query, the selectors that occur in no gate (the complex q_lookup of the one-advice-column
// shapes) first.  Pinned by the reference's known answer: for the k = 17 shape and the verifying-key commitments of
// the generated verifier the rendering hashes to the `transcript_repr` of proving-server/P256Verifier.yul:34
// (tests/test_oracle_kat.py checks the oracle's restatement against it; tests/test_gpu_prover.py checks this one
// against the oracle's).
//
// The engine indexes fixed columns in QUERY order (constants, table, selectors: the order of the fixed evaluations
// in a proof); halo2's column index of internal column i is halo2_fixed_column(lay, i).
#pragma once
#include <stdio.h>

#include <string>

#include "pk.h"
#include "transcript.h"

namespace vkrepr {

inline bool supported(const Layout&) { return true; }  // (never-enabled gate columns used to be excluded)

inline uint32_t halo2_fixed_column(const Layout& lay, uint32_t i) { return i < lay.F ? i + 1 : (i == lay.F ? 0 : i); }

// internal fixed-column indices in halo2's column order (fixed_commitments, VerifyingKey / ProvingKey files)
inline std::vector<uint32_t> halo2_fixed_order(const Layout& lay) {
    std::vector<uint32_t> order(lay.n_fix);
    for (uint32_t i = 0; i < lay.n_fix; i++) order[halo2_fixed_column(lay, i)] = i;
    return order;
}

template <class F>
inline std::string hex_of(const F& mont) {  // halo2curves Debug of a field element: 0x + 64 hex digits, big-endian
    const F c = fe_from_mont(mont);
    char buf[67];
    char* p = buf;
    p += snprintf(p, 3, "0x");
    for (int w = 7; w >= 0; w--) p += snprintf(p, 9, "%08x", c.v[w]);
    return std::string(buf);
}
inline std::string small_fe(uint32_t v) {  // the same Debug form of a small constant (Expression::Constant)
    char buf[67];
    snprintf(buf, sizeof(buf), "0x%056x%08x", 0u, v);
    return std::string(buf);
}
inline std::string point(const G1Affine& a) {
    if (affine_is_identity(a)) return "Infinity";
    return "(" + hex_of(a.x) + ", " + hex_of(a.y) + ")";
}
inline std::string column(uint32_t i, const char* type) {
    return "Column { index: " + std::to_string(i) + ", column_type: " + type + " }";
}
inline std::string advice(uint32_t qi, uint32_t ci, int rot) {
    return "Advice { query_index: " + std::to_string(qi) + ", column_index: " + std::to_string(ci) + ", rotation: Rotation(" +
           std::to_string(rot) + ") }";
}
inline std::string fixed(const Layout& lay, uint32_t i) {
    return "Fixed { query_index: " + std::to_string(i) + ", column_index: " + std::to_string(halo2_fixed_column(lay, i)) +
           ", rotation: Rotation(0) }";
}
template <class It, class Fn>
inline std::string join(It begin, It end, Fn fn) {
    std::string s;
    for (It it = begin; it != end; ++it) {
        if (it != begin) s += ", ";
        s += fn(*it);
    }
    return s;
}

// format!("{:?}", vk.pinned())
inline std::string pinned_debug(const Layout& lay, const std::vector<G1Affine>& fixed_commit, const std::vector<G1Affine>& perm_commit) {
    std::vector<uint32_t> gate_cols(lay.A), fix_idx(lay.n_fix), lk_idx(lay.n_lookup_cols);
    for (uint32_t j = 0; j < lay.A; j++) gate_cols[j] = j;
    for (uint32_t i = 0; i < lay.n_fix; i++) fix_idx[i] = i;
    for (uint32_t l = 0; l < lay.n_lookup_cols; l++) lk_idx[l] = l;
    const std::string gates = join(gate_cols.begin(), gate_cols.end(), [&](uint32_t j) {
        // the selector after compress_selectors: q, or for a combined pair q * (other_root - q) (pk.h Layout::gate_sel)
        const uint32_t form = lay.gate_sel[j] >> 24;
        std::string q = fixed(lay, lay.gate_sel[j] & 0xffffffu);
        if (form) q = "Product(" + q + ", Sum(Constant(" + small_fe(form == 1 ? 2 : 1) + "), Negated(" + q + ")))";
        return "Product(" + q + ", Sum(Sum(" + advice(4 * j, j, 0) + ", Product(" + advice(4 * j + 1, j, 1) + ", " +
               advice(4 * j + 2, j, 2) + ")), Negated(" + advice(4 * j + 3, j, 3) + ")))";
    });
    const std::string advice_queries = join(lay.advice_queries.begin(), lay.advice_queries.end(), [&](const std::pair<uint32_t, int>& q) {
        return "(" + column(q.first, "Advice") + ", Rotation(" + std::to_string(q.second) + "))";
    });
    const std::string fixed_queries = join(fix_idx.begin(), fix_idx.end(), [&](uint32_t i) {
        return "(" + column(halo2_fixed_column(lay, i), "Fixed") + ", Rotation(0))";
    });
    const std::string perm_cols = join(lay.perm_cols.begin(), lay.perm_cols.end(), [&](const Col& c) {
        return c.fixed ? column(halo2_fixed_column(lay, c.idx), "Fixed") : column(c.idx, "Advice");
    });
    const std::string table = fixed(lay, lay.fx_table);
    std::string lookups;
    uint32_t num_selectors;
    if (lay.single) {
        lookups = "Argument { input_expressions: [Product(" + fixed(lay, lay.fx_qlookup) + ", " + advice(0, 0, 0) + ")], table_expressions: [" +
                  table + "] }";
        num_selectors = 2;
    } else {
        lookups = join(lk_idx.begin(), lk_idx.end(), [&](uint32_t l) {
            return "Argument { input_expressions: [" + advice(4 * lay.A + l, lay.A + l, 0) + "], table_expressions: [" + table + "] }";
        });
        num_selectors = lay.A;
    }
    const std::vector<uint32_t> order = halo2_fixed_order(lay);
    std::string s = "PinnedVerificationKey { base_modulus: \"0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47\", "
                    "scalar_modulus: \"0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001\", domain: PinnedEvaluationDomain { k: " +
                    std::to_string(lay.k) + ", extended_k: " + std::to_string(lay.ext_k) + ", omega: " + hex_of(fr_omega(lay.k)) + " }, ";
    s += "cs: PinnedConstraintSystem { num_fixed_columns: " + std::to_string(lay.n_fix) + ", num_advice_columns: " + std::to_string(lay.n_adv) +
         ", num_instance_columns: 0, num_selectors: " + std::to_string(num_selectors) + ", gates: [" + gates + "], advice_queries: [" +
         advice_queries + "], instance_queries: [], fixed_queries: [" + fixed_queries + "], permutation: Argument { columns: [" + perm_cols +
         "] }, lookups: [" + lookups + "], constants: [], minimum_degree: None }, ";
    s += "fixed_commitments: [" + join(order.begin(), order.end(), [&](uint32_t i) { return point(fixed_commit[i]); }) + "], ";
    s += "permutation: VerifyingKey { commitments: [" + join(perm_commit.begin(), perm_commit.end(), [&](const G1Affine& a) { return point(a); }) +
         "] } }";
    return s;
}

inline Fr transcript_repr(const Layout& lay, const std::vector<G1Affine>& fixed_commit, const std::vector<G1Affine>& perm_commit) {
    const std::string s = pinned_debug(lay, fixed_commit, perm_commit);
    Blake2b h("Halo2-Verify-Key");
    const uint64_t len = s.size();
    h.update((const uint8_t*)&len, 8);
    h.update((const uint8_t*)s.data(), s.size());
    uint8_t dg[64];
    h.finalize_copy(dg);
    return fr_from_u512_le(dg);
}

}  // namespace vkrepr
