// prove_host_phases — the RESIDENT integration of INTEGRATION.md §2 executed: a host that keeps halo2's own prover flow
// and off-loads phase by phase.
//
// What a patched halo2_proofs `create_proof` (call sites: halo2-circuits/src/ecc/ecdsa_p256.rs:366-373 — EvmTranscript +
// ProverGWC, the /prove_evm path — and :416-423) does with the engine when it does NOT hand the whole proof to zk_prove: the
// HOST owns the transcript, the RNG and the blinding — every `Fr::random`, every challenge and every byte of the proof are made
// here — and the DEVICE does the arithmetic between them through the phase-level C ABI (include/zkmi355.h):
//     commitments            zk_commit_batch                (ParamsKZG::commit_lagrange / commit)
//     lookup argument        zk_lookup_permute, zk_lookup_product
//     permutation argument   zk_permutation_product
//     vanishing argument     zk_random_poly (the host hands over its ChaCha20 key and position), zk_quotient, zk_extended_to_coeff
//     domain                 zk_lagrange_to_coeff, zk_coeff_to_extended
//     openings               zk_eval, zk_poly_lincomb, zk_kate_division   (ProverGWC and ProverSHPLONK)
// No column crosses PCIe after the advice upload: the blinding rows go up as 7-row ranges (zk_poly_upload_range).  The proof
// is byte-identical to zk_prove's with the same key, advice and seed (tests/test_gpu_host_phases.py) — here the seed feeds
// THIS file's ChaCha20Rng, drawn in halo2's order.
//
// The host-side field arithmetic, Keccak / Blake2b transcripts and ChaCha20 are taken from the engine's host headers (csrc/
// transcript.h, hostutil.h: plain C++ behind HIP's function attributes, hence hipcc) — a Rust host has halo2curves and
// halo2_proofs::transcript for that; nothing of the engine's prover (csrc/prover.hip) is used.
//
// usage: prove_host_phases <srs.bin> <pk.bin> <advice.bin> <proof.out> k num_advice num_lookup_advice num_fixed lookup_bits idle
//                          <transcript: blake2b|evm> <rng seed: 64 hex digits> [gwc|shplonk]
//   multi-open: the reference's pairings by default — GWC under the EVM transcript (ecdsa_p256.rs:368), SHPLONK under Blake2b (:418)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <utility>
#include <vector>

#include "../include/zkmi355.h"
#include "../webauthn-halo2_amd/csrc/transcript.h"

using namespace zk;

static std::vector<uint8_t> slurp(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) {
        fprintf(stderr, "prove_host_phases: cannot open %s\n", path);
        exit(2);
    }
    fseek(f, 0, SEEK_END);
    const long len = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> buf((size_t)len);
    if (len && fread(buf.data(), 1, (size_t)len, f) != (size_t)len) {
        fprintf(stderr, "prove_host_phases: short read of %s\n", path);
        exit(2);
    }
    fclose(f);
    return buf;
}

static zk_ctx* ctx = nullptr;
#define CHECK(call)                                                                                                      \
    do {                                                                                                                 \
        const int rc_ = (call);                                                                                          \
        if (rc_ != ZK_OK) {                                                                                              \
            fprintf(stderr, "prove_host_phases: %s failed: %d (%s)\n", #call, rc_, zk_strerror(rc_));                    \
            exit(1);                                                                                                     \
        }                                                                                                                \
    } while (0)

static const uint64_t* limbs(const Fr& a) { return reinterpret_cast<const uint64_t*>(a.v); }
static zk_poly alloc(size_t n) {
    zk_poly p = 0;
    CHECK(zk_poly_alloc(ctx, n, &p));
    return p;
}
static Fr eval(zk_poly p, const Fr& x) {
    Fr out;
    CHECK(zk_eval(ctx, p, limbs(x), reinterpret_cast<uint64_t*>(out.v)));
    return out;
}
// commits `polys` and writes the points to the transcript, in order
static void commit_write(Transcript& tr, const std::vector<zk_poly>& polys, int basis) {
    std::vector<G1Affine> pts(polys.size());
    CHECK(zk_commit_batch(ctx, polys.data(), polys.size(), basis, reinterpret_cast<uint64_t*>(pts.data())));
    for (const G1Affine& p : pts)
        if (!tr.write_point(p)) {
            fprintf(stderr, "prove_host_phases: a commitment is the identity\n");
            exit(1);
        }
}

int main(int argc, char** argv) {
    if (argc != 13 && argc != 14) {
        fprintf(stderr, "usage: %s srs.bin pk.bin advice.bin proof.out k A L F lookup_bits idle blake2b|evm seedhex [gwc|shplonk]\n", argv[0]);
        return 2;
    }
    zk_circuit_params prm;
    memset(&prm, 0, sizeof(prm));
    prm.k = (uint32_t)atoi(argv[5]);
    prm.num_advice = (uint32_t)atoi(argv[6]);
    prm.num_lookup_advice = (uint32_t)atoi(argv[7]);
    prm.num_fixed = (uint32_t)atoi(argv[8]);
    prm.lookup_bits = (uint32_t)atoi(argv[9]);
    prm.num_idle_gate_columns = (uint32_t)atoi(argv[10]);
    const bool evm = strcmp(argv[11], "evm") == 0;
    const bool shplonk = argc == 14 ? strcmp(argv[13], "shplonk") == 0 : !evm;
    uint8_t seed[32];
    if (strlen(argv[12]) != 64) return 2;
    for (int i = 0; i < 32; i++) {
        unsigned v;
        sscanf(argv[12] + 2 * i, "%2x", &v);
        seed[i] = (uint8_t)v;
    }
    CHECK(zk_ctx_create(0, &ctx));
    {
        const std::vector<uint8_t> srs = slurp(argv[1]);
        CHECK(zk_srs_read(ctx, srs.data(), srs.size(), ZK_SERDE_RAW_BYTES));
    }
    zk_pk pk = 0;
    {
        const std::vector<uint8_t> key = slurp(argv[2]);
        CHECK(zk_pk_read(ctx, &prm, key.data(), key.size(), ZK_SERDE_RAW_BYTES, nullptr, &pk));
    }
    // ---- what the host knows of the circuit: halo2-lib's column shape for this config row (ConstraintSystem of the key)
    uint32_t shape[8];
    CHECK(zk_pk_shape(ctx, pk, shape));
    const uint32_t k = shape[0], n_adv = shape[2], n_fix = shape[3], n_perm = shape[4], n_chunks = shape[5], n_lookups = shape[6], n_h = shape[7];
    const size_t n = (size_t)1 << k, N = 4 * n;
    const uint32_t A = prm.num_advice;         // gate columns: queried at rotations 0 .. 3; lookup advice columns at 0
    const uint32_t bf = 6;                     // blinding factors; rows n - 7 .. n - 1 of a column are the prover's
    const size_t usable = n - (bf + 1);
    const int last_rot = -(int)(bf + 1);
    Fr repr;
    {
        uint32_t counts[2];
        CHECK(zk_vk_export(ctx, pk, nullptr, nullptr, reinterpret_cast<uint64_t*>(repr.v), counts));
    }
    const Fr omega = fr_omega(k), omega_inv = fe_inv(omega);
    auto xrot = [&](const Fr& x, int r) { return fe_mul(x, fe_pow_u64(r >= 0 ? omega : omega_inv, (uint64_t)(r >= 0 ? r : -r))); };

    EvmTranscript evm_tr;
    Blake2bTranscript b2_tr;
    Transcript& tr = evm ? static_cast<Transcript&>(evm_tr) : static_cast<Transcript&>(b2_tr);
    ChaCha20Rng rng(seed);  // the host's RNG (the reference: OsRng)
    auto draw = [&](uint32_t count) {
        std::vector<Fr> v(count);
        for (Fr& x : v) x = rng.next_fr();
        return v;
    };
    auto blind = [&](zk_poly col, size_t first, uint32_t count) {  // rows [first, first + count) = fresh randomness
        const std::vector<Fr> v = draw(count);
        CHECK(zk_poly_upload_range(ctx, col, first, reinterpret_cast<const uint64_t*>(v.data()), count));
    };
    tr.common_scalar(repr);

    // ---- 1. advice: the host's own columns (ECDSACircuit::synthesize, ecdsa_p256.rs:117-206), blinded on the host, shipped once
    std::vector<zk_poly> adv(n_adv);
    {
        const std::vector<uint8_t> file = slurp(argv[3]);
        if (file.size() != (size_t)n_adv * n * 32) {
            fprintf(stderr, "prove_host_phases: %s does not hold %u columns of %zu rows\n", argv[3], n_adv, n);
            return 2;
        }
        std::vector<Fr> col(n);
        for (uint32_t j = 0; j < n_adv; j++) {
            for (size_t r = 0; r < usable; r++) {
                Fr c;
                memcpy(c.v, file.data() + ((size_t)j * n + r) * 32, 32);
                col[r] = fe_to_mont(c);
            }
            const std::vector<Fr> b = draw(bf + 1);
            for (uint32_t t = 0; t <= bf; t++) col[usable + t] = b[t];
            adv[j] = alloc(n);
            CHECK(zk_poly_upload(ctx, adv[j], reinterpret_cast<const uint64_t*>(col.data()), n));
        }
        draw(n_adv);  // the advice blinds (Blind::default is drawn even though KZG ignores it)
    }
    commit_write(tr, adv, ZK_BASIS_LAGRANGE);
    (void)tr.squeeze();  // theta: every lookup of this circuit family is a single expression

    // ---- 2. lookup argument: permuted input / table on the device, their blinding here
    std::vector<zk_poly> ap(n_lookups), sp(n_lookups), zl(n_lookups);
    for (uint32_t l = 0; l < n_lookups; l++) {
        ap[l] = alloc(n);
        sp[l] = alloc(n);
        zl[l] = alloc(n);
    }
    CHECK(zk_lookup_permute(ctx, pk, adv.data(), n_adv, ap.data(), sp.data(), n_lookups));
    {
        std::vector<zk_poly> order;
        for (uint32_t l = 0; l < n_lookups; l++) {
            blind(ap[l], usable, bf + 1);
            blind(sp[l], usable, bf + 1);
            draw(2);
            order.push_back(ap[l]);
            order.push_back(sp[l]);
        }
        commit_write(tr, order, ZK_BASIS_LAGRANGE);
    }
    const Fr beta = tr.squeeze(), gamma = tr.squeeze();

    // ---- 3. grand products: permutation chunks, then lookups
    std::vector<zk_poly> z(n_chunks);
    for (uint32_t ci = 0; ci < n_chunks; ci++) z[ci] = alloc(n);
    CHECK(zk_permutation_product(ctx, pk, adv.data(), n_adv, limbs(beta), limbs(gamma), z.data(), n_chunks));
    CHECK(zk_lookup_product(ctx, pk, adv.data(), n_adv, ap.data(), sp.data(), n_lookups, limbs(beta), limbs(gamma), zl.data()));
    {
        std::vector<zk_poly> order;
        for (uint32_t ci = 0; ci < n_chunks; ci++) {
            blind(z[ci], n - bf, bf);
            draw(1);
            order.push_back(z[ci]);
        }
        for (uint32_t l = 0; l < n_lookups; l++) {
            blind(zl[l], n - bf, bf);
            draw(1);
            order.push_back(zl[l]);
        }
        commit_write(tr, order, ZK_BASIS_LAGRANGE);
    }

    // ---- 4. vanishing argument: the random polynomial — n draws of the host's stream, expanded on the device
    zk_poly rnd = alloc(n);
    CHECK(zk_random_poly(ctx, rng.key, rng.block, rnd));
    rng.block += n;
    draw(1);
    commit_write(tr, {rnd}, ZK_BASIS_MONOMIAL);
    const Fr y = tr.squeeze();

    // ---- 5. coefficient and extended-coset forms of every committed column, then h(X)
    auto to_poly = [&](zk_poly values) {
        zk_poly p = alloc(n);
        CHECK(zk_poly_copy(ctx, p, values));
        CHECK(zk_lagrange_to_coeff(ctx, p));
        return p;
    };
    auto to_ext = [&](zk_poly poly) {
        zk_poly e = alloc(N);
        CHECK(zk_coeff_to_extended(ctx, poly, e));
        return e;
    };
    std::vector<zk_poly> adv_p(n_adv), adv_e(n_adv), z_p(n_chunks), z_e(n_chunks), ap_p(n_lookups), sp_p(n_lookups), zl_p(n_lookups), lk_e;
    for (uint32_t j = 0; j < n_adv; j++) adv_e[j] = to_ext(adv_p[j] = to_poly(adv[j]));
    for (uint32_t ci = 0; ci < n_chunks; ci++) z_e[ci] = to_ext(z_p[ci] = to_poly(z[ci]));
    for (uint32_t l = 0; l < n_lookups; l++) {
        lk_e.push_back(to_ext(ap_p[l] = to_poly(ap[l])));
        lk_e.push_back(to_ext(sp_p[l] = to_poly(sp[l])));
        lk_e.push_back(to_ext(zl_p[l] = to_poly(zl[l])));
    }
    zk_poly h_ext = alloc(N);
    CHECK(zk_quotient(ctx, pk, adv_e.data(), n_adv, z_e.data(), n_chunks, lk_e.data(), n_lookups, limbs(beta), limbs(gamma), limbs(y), 1, h_ext));
    CHECK(zk_extended_to_coeff(ctx, h_ext, (size_t)n_h * n));
    draw(n_h);  // the h pieces' blinds
    std::vector<zk_poly> h_piece(n_h);
    for (uint32_t i = 0; i < n_h; i++) {
        h_piece[i] = alloc(n);
        CHECK(zk_poly_copy_range(ctx, h_piece[i], 0, h_ext, (size_t)i * n, n));
    }
    commit_write(tr, h_piece, ZK_BASIS_MONOMIAL);
    const Fr x = tr.squeeze();

    // ---- 6. evaluations, in transcript order (plonk::prover: advice, fixed, vanishing, permutation, lookups)
    std::vector<zk_poly> fixed_p(n_fix), sigma_p(n_perm);
    for (uint32_t f = 0; f < n_fix; f++) CHECK(zk_pk_export_poly(ctx, pk, ZK_PK_FIXED_POLY, f, fixed_p[f] = alloc(n)));
    for (uint32_t p = 0; p < n_perm; p++) CHECK(zk_pk_export_poly(ctx, pk, ZK_PK_SIGMA_POLY, p, sigma_p[p] = alloc(n)));
    zk_poly h_comb = alloc(n);
    {
        std::vector<Fr> c(n_h);
        const Fr xn = fe_pow_u64(x, n);
        Fr p = Fr::one();
        for (uint32_t i = 0; i < n_h; i++) {
            c[i] = p;
            p = fe_mul(p, xn);
        }
        CHECK(zk_poly_lincomb(ctx, h_comb, h_piece.data(), reinterpret_cast<const uint64_t*>(c.data()), n_h, nullptr, 0));
    }
    struct Q {
        zk_poly poly;
        int rot;
        Fr eval;
    };
    std::vector<Q> ev;
    for (uint32_t j = 0; j < A; j++)
        for (int r = 0; r < 4; r++) ev.push_back(Q{adv_p[j], r, Fr::zero()});
    for (uint32_t j = A; j < n_adv; j++) ev.push_back(Q{adv_p[j], 0, Fr::zero()});
    const size_t i_fix = ev.size();
    for (uint32_t f = 0; f < n_fix; f++) ev.push_back(Q{fixed_p[f], 0, Fr::zero()});
    const size_t i_rand = ev.size();
    ev.push_back(Q{rnd, 0, Fr::zero()});
    const size_t i_sig = ev.size();
    for (uint32_t p = 0; p < n_perm; p++) ev.push_back(Q{sigma_p[p], 0, Fr::zero()});
    const size_t i_z = ev.size();
    for (uint32_t ci = 0; ci < n_chunks; ci++) {
        ev.push_back(Q{z_p[ci], 0, Fr::zero()});
        ev.push_back(Q{z_p[ci], 1, Fr::zero()});
        if (ci != n_chunks - 1) ev.push_back(Q{z_p[ci], last_rot, Fr::zero()});
    }
    const size_t i_lk = ev.size();
    for (uint32_t l = 0; l < n_lookups; l++) {
        ev.push_back(Q{zl_p[l], 0, Fr::zero()});
        ev.push_back(Q{zl_p[l], 1, Fr::zero()});
        ev.push_back(Q{ap_p[l], 0, Fr::zero()});
        ev.push_back(Q{ap_p[l], -1, Fr::zero()});
        ev.push_back(Q{sp_p[l], 0, Fr::zero()});
    }
    const size_t n_written = ev.size();
    ev.push_back(Q{h_comb, 0, Fr::zero()});
    for (Q& q : ev) q.eval = eval(q.poly, xrot(x, q.rot));
    for (size_t i = 0; i < n_written; i++) tr.write_scalar(ev[i].eval);

    // ---- 7. multi-open (ProverGWC): queries in the prover's order, one witness polynomial per distinct rotation
    std::vector<Q> queries(ev.begin(), ev.begin() + i_fix);
    {
        std::vector<Q> lastq(n_chunks);
        size_t pos = i_z;
        for (uint32_t ci = 0; ci < n_chunks; ci++) {
            queries.push_back(ev[pos++]);
            queries.push_back(ev[pos++]);
            if (ci != n_chunks - 1) lastq[ci] = ev[pos++];
        }
        for (int ci = (int)n_chunks - 2; ci >= 0; ci--) queries.push_back(lastq[ci]);
        pos = i_lk;
        for (uint32_t l = 0; l < n_lookups; l++, pos += 5) {
            queries.push_back(ev[pos]);      // zL @ x
            queries.push_back(ev[pos + 2]);  // a' @ x
            queries.push_back(ev[pos + 4]);  // s' @ x
            queries.push_back(ev[pos + 3]);  // a' @ w^-1 x
            queries.push_back(ev[pos + 1]);  // zL @ w x
        }
        for (size_t i = i_fix; i < i_rand; i++) queries.push_back(ev[i]);
        for (size_t i = i_sig; i < i_z; i++) queries.push_back(ev[i]);
        queries.push_back(ev[n_written]);  // h
        queries.push_back(ev[i_rand]);     // the random polynomial
    }
    if (!shplonk) {
        const Fr v = tr.squeeze();
        std::vector<std::pair<int, std::vector<Q>>> sets;
        for (const Q& q : queries) {
            bool found = false;
            for (auto& s : sets)
                if (s.first == q.rot) {
                    s.second.push_back(q);
                    found = true;
                    break;
                }
            if (!found) sets.push_back({q.rot, {q}});
        }
        std::vector<zk_poly> wit;
        for (auto& s : sets) {
            // (sum_i v^i p_i(X) - sum_i v^i e_i) / (X - x w^rot).  A polynomial opened twice in one set (none here) would appear twice.
            std::vector<zk_poly> in;
            std::vector<Fr> c;
            Fr pv = Fr::one(), eb = Fr::zero();
            for (const Q& q : s.second) {
                in.push_back(q.poly);
                c.push_back(pv);
                eb = fe_add(eb, fe_mul(pv, q.eval));
                pv = fe_mul(pv, v);
            }
            zk_poly w = alloc(n);
            CHECK(zk_poly_lincomb(ctx, w, in.data(), reinterpret_cast<const uint64_t*>(c.data()), in.size(), limbs(eb), 1));
            const Fr pt = xrot(x, s.first);
            CHECK(zk_kate_division(ctx, w, limbs(pt), w));
            wit.push_back(w);
        }
        commit_write(tr, wit, ZK_BASIS_MONOMIAL);
    } else {
        // ---- 7'. multi-open (ProverSHPLONK): polynomials grouped by their set of rotations; h(X) = sum_i v^i (sum_j y^j (P_ij - R_ij)) / Z_i
        struct CR {
            zk_poly poly;
            std::vector<int> rots;
            std::vector<Fr> evals;
        };
        std::vector<CR> com;
        for (const Q& q : queries) {
            size_t at = com.size();
            for (size_t i = 0; i < com.size(); i++)
                if (com[i].poly == q.poly) at = i;
            if (at == com.size()) com.push_back(CR{q.poly, {}, {}});
            com[at].rots.push_back(q.rot);
            com[at].evals.push_back(q.eval);
        }
        auto canon_less = [&](int ra, int rb) {  // BTreeSet<Fr>: the points ordered by their canonical integer value
            const Fr a = fe_from_mont(xrot(x, ra)), b = fe_from_mont(xrot(x, rb));
            for (int i = 7; i >= 0; i--)
                if (a.v[i] != b.v[i]) return a.v[i] < b.v[i];
            return false;
        };
        struct RS {
            std::vector<int> rots;
            std::vector<size_t> coms;
        };
        std::vector<RS> rsets;
        std::vector<int> all_rots;
        for (size_t ci = 0; ci < com.size(); ci++) {
            CR& cr = com[ci];
            std::vector<size_t> order(cr.rots.size());
            for (size_t i = 0; i < order.size(); i++) order[i] = i;
            std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return canon_less(cr.rots[a], cr.rots[b]); });
            std::vector<int> r2;
            std::vector<Fr> e2;
            for (size_t i : order) {
                r2.push_back(cr.rots[i]);
                e2.push_back(cr.evals[i]);
            }
            cr.rots = r2;
            cr.evals = e2;
            for (int r : cr.rots)
                if (std::find(all_rots.begin(), all_rots.end(), r) == all_rots.end()) all_rots.push_back(r);
            size_t hit = rsets.size();
            for (size_t si = 0; si < rsets.size(); si++)
                if (rsets[si].rots == cr.rots) hit = si;
            if (hit == rsets.size()) rsets.push_back(RS{cr.rots, {}});
            rsets[hit].coms.push_back(ci);
        }
        std::sort(all_rots.begin(), all_rots.end(), canon_less);
        const Fr yc = tr.squeeze(), v = tr.squeeze();
        // the Lagrange basis over a set's points as coefficient vectors, one inversion for all denominators (lagrange_interpolate)
        auto lagrange_basis = [&](const std::vector<Fr>& pts) {
            const size_t m = pts.size();
            std::vector<std::vector<Fr>> basis(m);
            std::vector<Fr> den(m, Fr::one());
            for (size_t j = 0; j < m; j++) {
                std::vector<Fr> num(1, Fr::one());
                for (size_t i = 0; i < m; i++) {
                    if (i == j) continue;
                    std::vector<Fr> nn(num.size() + 1, Fr::zero());
                    for (size_t t = 0; t < num.size(); t++) {
                        nn[t + 1] = fe_add(nn[t + 1], num[t]);
                        nn[t] = fe_sub(nn[t], fe_mul(pts[i], num[t]));
                    }
                    num.swap(nn);
                    den[j] = fe_mul(den[j], fe_sub(pts[j], pts[i]));
                }
                basis[j] = num;
            }
            for (size_t j = 0; j < m; j++) {
                const Fr dj = fe_inv(den[j]);
                for (Fr& cf : basis[j]) cf = fe_mul(cf, dj);
            }
            return basis;
        };
        std::vector<std::vector<Fr>> low(com.size());  // every polynomial's remainder over its set's points
        std::vector<zk_poly> sbuf;
        for (RS& rs : rsets) {
            std::vector<Fr> pts;
            for (int r : rs.rots) pts.push_back(xrot(x, r));
            const std::vector<std::vector<Fr>> basis = lagrange_basis(pts);
            std::vector<zk_poly> in;
            std::vector<Fr> c, rsum(pts.size(), Fr::zero());
            Fr py = Fr::one();
            for (size_t ci : rs.coms) {
                std::vector<Fr>& lo = low[ci];
                lo.assign(pts.size(), Fr::zero());
                for (size_t j = 0; j < pts.size(); j++)
                    for (size_t t = 0; t < pts.size(); t++) lo[t] = fe_add(lo[t], fe_mul(basis[j][t], com[ci].evals[j]));
                in.push_back(com[ci].poly);
                c.push_back(py);
                for (size_t t = 0; t < pts.size(); t++) rsum[t] = fe_add(rsum[t], fe_mul(py, lo[t]));
                py = fe_mul(py, yc);
            }
            zk_poly sb = alloc(n);
            CHECK(zk_poly_lincomb(ctx, sb, in.data(), reinterpret_cast<const uint64_t*>(c.data()), in.size(),
                                  reinterpret_cast<const uint64_t*>(rsum.data()), rsum.size()));
            for (const Fr& pt : pts) CHECK(zk_kate_division(ctx, sb, limbs(pt), sb));  // exact: the numerator vanishes on the set
            sbuf.push_back(sb);
        }
        zk_poly hx = alloc(n);
        {
            std::vector<Fr> c;
            Fr pv = Fr::one();
            for (size_t si = 0; si < rsets.size(); si++) {
                c.push_back(pv);
                pv = fe_mul(pv, v);
            }
            CHECK(zk_poly_lincomb(ctx, hx, sbuf.data(), reinterpret_cast<const uint64_t*>(c.data()), sbuf.size(), nullptr, 0));
        }
        commit_write(tr, {hx}, ZK_BASIS_MONOMIAL);
        const Fr u = tr.squeeze();
        // L(X) = sum_i v^i z_i(u) sum_j y^j (P_ij(X) - R_ij(u)) - Z_T(u) h(X);  the proof's last point commits to L(X) / ((X - u) z_0(u))
        auto vanishing_eval = [&](const std::vector<Fr>& pts, const Fr& at) {
            Fr acc = Fr::one();
            for (const Fr& p : pts) acc = fe_mul(acc, fe_sub(at, p));
            return acc;
        };
        std::vector<zk_poly> in;
        std::vector<Fr> c, z_diffs;
        Fr sub = Fr::zero(), pv = Fr::one();
        for (RS& rs : rsets) {
            std::vector<Fr> diffs;
            for (int r : all_rots)
                if (std::find(rs.rots.begin(), rs.rots.end(), r) == rs.rots.end()) diffs.push_back(xrot(x, r));
            const Fr zi = vanishing_eval(diffs, u);
            z_diffs.push_back(zi);
            Fr py = Fr::one();
            for (size_t ci : rs.coms) {
                const Fr coef = fe_mul(fe_mul(pv, zi), py);
                in.push_back(com[ci].poly);
                c.push_back(coef);
                Fr ru = Fr::zero();  // R_ij(u)
                for (size_t t = low[ci].size(); t-- > 0;) ru = fe_add(fe_mul(ru, u), low[ci][t]);
                sub = fe_add(sub, fe_mul(coef, ru));
                py = fe_mul(py, yc);
            }
            pv = fe_mul(pv, v);
        }
        std::vector<Fr> all_pts;
        for (int r : all_rots) all_pts.push_back(xrot(x, r));
        in.push_back(hx);
        c.push_back(fe_neg(vanishing_eval(all_pts, u)));
        zk_poly lx = alloc(n), fin = alloc(n);
        CHECK(zk_poly_lincomb(ctx, lx, in.data(), reinterpret_cast<const uint64_t*>(c.data()), in.size(), limbs(sub), 1));
        CHECK(zk_kate_division(ctx, lx, limbs(u), lx));
        const Fr zinv = fe_inv(z_diffs[0]);
        CHECK(zk_poly_lincomb(ctx, fin, &lx, limbs(zinv), 1, nullptr, 0));
        commit_write(tr, {fin}, ZK_BASIS_MONOMIAL);
    }

    FILE* out = fopen(argv[4], "wb");
    if (!out || fwrite(tr.out.data(), 1, tr.out.size(), out) != tr.out.size()) {
        fprintf(stderr, "prove_host_phases: cannot write %s\n", argv[4]);
        return 2;
    }
    fclose(out);
    printf("prove_host_phases: %zu proof bytes\n", tr.out.size());
    zk_pk_free(ctx, pk);
    zk_ctx_destroy(ctx);  // (frees every vector of the context)
    return 0;
}
