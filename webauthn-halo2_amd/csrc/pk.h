// pk.h — the resident proving key record shared by prover.hip (keygen / create_proof) and serde.hip
// (ProvingKey / VerifyingKey read & write).  Not part of the public C-ABI.
#pragma once
#include <utility>
#include <vector>

#include "ctx.h"
#include "prover.h"

struct Col {
    int fixed;  // 1 = fixed column, 0 = advice
    uint32_t idx;
};

struct Layout {
    uint32_t k, n, A, L, F, lookup_bits, idle;
    bool single;
    uint32_t n_gate, n_lookup_cols, n_adv, fx_table, fx_qlookup, n_fix;
    std::vector<uint32_t> fx_sel;    // the selector column a gate column OWNS (NO_SELECTOR: a never-enabled selector has none)
    // gate j's selector after halo2's compress_selectors: fixed column | form << 24.  A never-enabled simple selector excludes
    // nobody, so the greedy pass puts the t-th one into the column of gate t (degree 2 + 2 members <= 4), and both selectors of
    // the pair are replaced: form 1 = q (2 - q) (the used one), form 2 = q (1 - q) (the never-enabled one); form 0 = q
    std::vector<uint32_t> gate_sel;
    std::vector<Col> perm_cols;
    uint32_t n_lookups, degree, chunk_len, n_chunks, n_h, ext_k, usable;
    int last_rot;
    std::vector<std::pair<uint32_t, int>> advice_queries;

    bool init(const zk_circuit_params& p) {
        k = p.k; A = p.num_advice; L = p.num_lookup_advice; F = p.num_fixed; lookup_bits = p.lookup_bits;
        idle = p.num_idle_gate_columns;
        if (k < 4 || k > 22 || A < 1 || L < 1 || F < 1 || idle >= A || 2 * idle > A) return false;  // (more idle than used: pairs of their own)
        if (lookup_bits < 1 || lookup_bits >= k) return false;  // the range table 0 .. 2^lookup_bits - 1 must fit in the usable rows
        n = 1u << k;
        single = A == 1;
        n_gate = A;
        n_lookup_cols = single ? 0 : L;
        n_adv = A + n_lookup_cols;
        fx_table = F;
        fx_sel.clear();
        if (single) {
            // halo2's compress_selectors gives the selectors that occur in no gate (the complex q_lookup) their fixed columns
            // first, then the simple ones (vkrepr.h)
            fx_qlookup = F + 1;
            fx_sel.push_back(F + 2);
            n_fix = F + 3;
        } else {
            for (uint32_t j = 0; j < A; j++) fx_sel.push_back(j < A - idle ? F + 1 + j : NO_SELECTOR);
            fx_qlookup = 0;
            n_fix = F + 1 + A - idle;
        }
        gate_sel.assign(fx_sel.begin(), fx_sel.end());
        for (uint32_t t = 0; t < (single ? 0u : idle); t++) {
            gate_sel[t] = fx_sel[t] | (1u << 24);
            gate_sel[A - idle + t] = fx_sel[t] | (2u << 24);
        }
        perm_cols.clear();
        for (uint32_t f = 0; f < F; f++) perm_cols.push_back(Col{1, f});
        for (uint32_t j = 0; j < n_adv; j++) perm_cols.push_back(Col{0, j});
        n_lookups = single ? 1 : L;
        degree = single ? 5 : 4;
        chunk_len = degree - 2;
        n_chunks = ((uint32_t)perm_cols.size() + chunk_len - 1) / chunk_len;
        n_h = degree - 1;
        ext_k = k + 2;
        usable = n - (BLINDING_FACTORS + 1);
        last_rot = -(int)(BLINDING_FACTORS + 1);
        advice_queries.clear();
        for (uint32_t j = 0; j < A; j++)
            for (int r = 0; r < 4; r++) advice_queries.push_back({j, r});
        for (uint32_t l = 0; l < n_lookup_cols; l++) advice_queries.push_back({A + l, 0});
        if ((1u << lookup_bits) >= usable) return false;
        return n_adv <= MAX_ADV && n_fix <= MAX_FIX && perm_cols.size() <= MAX_PERM && n_chunks <= MAX_CHUNKS &&
               n_lookups <= MAX_LOOKUPS;
    }
};

// Whether the closed form above (Layout::gate_sel) is what halo2's compress_selectors would build from these selector
// activations (bits[j]: selector j's 2^k rows, LSB-first, the VerifyingKey::write packing; j < n_gate).  The greedy pass
// combines simple selectors that are never enabled on a common row, so the closed form — one column per USED selector,
// never-enabled selector t in the column of gate t — holds iff (a) every selector declared used is enabled somewhere and
// every pair of used selectors shares a row (no two of them are combined with each other), and (b) the selectors declared
// idle are all-zero.  Anything else would make halo2 pair columns differently: another vk digest, other gates — refused
// with ZK_ELAYOUT instead of proofs that silently diverge.  (2 * idle <= A is Layout::init's own limit: more never-enabled
// selectors than used ones would pair up in all-zero columns of their own.)
inline bool layout_selectors_fit(const Layout& lay, const std::vector<std::vector<uint8_t>>& bits) {
    if (lay.single) return true;  // one simple selector; the complex q_lookup is never combined
    const uint32_t used = lay.A - lay.idle, bytes = lay.n / 8;
    if (bits.size() < lay.A) return false;
    for (uint32_t j = used; j < lay.A; j++)
        for (uint32_t i = 0; i < bytes; i++)
            if (bits[j][i]) return false;
    for (uint32_t a = 0; a < used; a++) {
        for (uint32_t b = a; b < used; b++) {  // b == a: enabled somewhere
            bool share = false;
            for (uint32_t i = 0; i < bytes && !share; i++) share = (bits[a][i] & bits[b][i]) != 0;
            if (!share) return false;
        }
    }
    return true;
}

static constexpr uint32_t BATCH_ARGS_MIN = 8;  // more chunks / lookups / columns than this: one batched launch, argument blocks in device memory
static constexpr uint32_t ROWS_CAP = 512, ROWS_BLOCKS = 16;  // staged row writes per flush / flushes per ring

struct zk_pk_rec {
    Layout lay;
    uint64_t srs_gen = 0;  // the context's SRS generation the key's commitments belong to
    std::vector<Fr*> dev;  // every device allocation (freed together)
    std::vector<Fr*> fixed_val, fixed_poly, fixed_coset, sigma_val, sigma_poly, sigma_coset;
    Fr *l0_coset = nullptr, *l_last_coset = nullptr, *l_active_coset = nullptr;
    std::vector<Fr*> fixed_c3, sigma_c3;  // pk_ensure_cosets3 (empty until the first proof on the three-coset route)
    Fr *l0_c3 = nullptr, *l_last_c3 = nullptr, *l_active_c3 = nullptr;
    std::vector<G1Affine> fixed_commit, perm_commit;
    Fr transcript_repr;
    // prover workspace
    std::vector<Fr*> adv_val, adv_poly, adv_coset;
    std::vector<Fr*> z_val, z_poly, z_coset;
    std::vector<Fr*> lk_in, lk_ap, lk_ap_poly, lk_ap_coset, lk_sp, lk_sp_poly, lk_sp_coset, lk_z, lk_z_poly, lk_z_coset,
        lk_in_coset;
    Fr *random_poly = nullptr, *h_ext = nullptr, *h_comb = nullptr;
    Fr *t_num = nullptr, *t_den = nullptr, *t_frac = nullptr, *t_a = nullptr, *t_b = nullptr, *t_small = nullptr;
    Fr* kd_scratch = nullptr;  // Kate divisions: KD_MAX_BATCH x kate_division_scratch(n)
    Fr* tail_host = nullptr;  // pinned staging for evaluations / scalars
    RowEntry* rows_host = nullptr;  // pinned: ROWS_BLOCKS blocks of ROWS_CAP staged row writes (Prover::set_rows)
    RowEntry* rows_dev = nullptr;
    LookupScratch lks{};
    uint32_t* lk_u32 = nullptr;
    // all grand products of a proof in one batch (launch_gp_batch_*)
    std::vector<Fr*> gp_num, gp_den, gp_loc_p, gp_loc_r;  // per product, n each
    Fr* gp_tot = nullptr;        // 2 x blocks per product
    Fr* gp_scal = nullptr;       // device: q, q_inv, k, init (n_prod each)
    Fr* gp_host = nullptr;       // pinned: q and q_inv
    GpItem* d_gp_items = nullptr;
    // argument blocks of the batched per-chunk / per-lookup / per-column launches (many-column shapes): pinned staging + device
    void *h_batch_args = nullptr, *d_batch_args = nullptr;
    size_t batch_args_bytes = 0;
    QuotientArgs* d_qargs = nullptr;
    QuotientArgs* h_qargs = nullptr;  // pinned staging of the same
    EvalItem *d_evargs = nullptr, *h_evargs = nullptr;
    Fr t_inv[4];               // 1 / ((zeta w_ext^i)^n - 1), i < 4 (pk_quotient)
    bool t_inv_ready = false;
    uint32_t max_evals = 0;
    LcTerm *d_lc_terms = nullptr, *h_lc_terms = nullptr;  // argument lists of the multi-open's long linear combinations
    uint32_t lc_cap = 0, lc_used = 0;                     // slots, and how many this proof has used so far
    Fr *ev_scratch = nullptr, *ev_out = nullptr;
    // ---- lock-step batches (zk_prove_batch, prover_batch.h): proof j > 0 of a batch works in members[j - 1], a record whose
    // key half (fixed / sigma / l_* vectors, commitments, transcript_repr) ALIASES this key's and whose workspace is its own
    // (`dev` of a member lists its workspace only); `bb` holds what the merged launches of a batch need across proofs
    bool is_member = false;
    std::vector<zk_pk_rec*> members;
    struct BatchBufs* bb = nullptr;
};

// buffers of the launches a lock-step batch shares between its proofs, sized for `cap` proofs
struct BatchBufs {
    uint32_t cap = 0;
    uint32_t* lk_u32 = nullptr;   // lookup scratch of cap x n_lookups lookups (LookupScratch layout)
    LookupScratch lks{};
    GpItem* d_gp_items = nullptr; // cap x (chunks + lookups) grand products in one batch
    Fr* gp_scal = nullptr;        // device: q, q_inv, k, init
    Fr* gp_host = nullptr;        // pinned: q and q_inv
    EvalItem *d_evargs = nullptr, *h_evargs = nullptr;  // every opened value of every proof in one launch
    Fr *ev_scratch = nullptr, *ev_out = nullptr, *tail_host = nullptr;
};


// every device allocation of a key goes through Dev (freed together by pk_destroy)
struct Dev {
    zk_ctx* c;
    zk_pk_rec* pk;
    int rc = ZK_OK;

    Fr* alloc(size_t n) {
        Fr* p = nullptr;
        if (rc) return nullptr;
        if (hipMalloc(&p, n * sizeof(Fr)) != hipSuccess) {
            rc = ZK_ENOMEM;
            return nullptr;
        }
        pk->dev.push_back(p);
        return p;
    }
};

// extended cosets the quotient reads beside the key's own: advice columns, permutation products, per lookup a', s', zL
struct QuotientCosets {
    std::vector<const Fr*> adv, z, lk_a, lk_s, lk_z;
    bool cosets3 = false;  // the operands are [3][n] coset-major vectors (poly.hip "three cosets"); the key's own are taken from its c3 copies
};
// the key's extended cosets (fixed, sigma, l_0, l_last, l_active) once more in the [3][n] coset-major order of the three-coset
// route: made on the first proof that takes it (keygen and zk_pk_read both end up here), kept with the key
int pk_ensure_cosets3(zk_ctx* c, zk_pk_rec* pk);
int pk_quotient(zk_ctx* c, zk_pk_rec* pk, const QuotientCosets& qc, const Fr& beta, const Fr& gamma, const Fr& y, bool divide, Fr* out);
void pk_destroy(zk_pk_rec* pk);
// the per-proof workspace (advice / z / lookup forms, quotient buffer, scan and evaluation scratch): everything a key
// needs beyond the key material itself; called at the end of zk_keygen and zk_pk_read
int pk_alloc_workspace(zk_ctx* c, zk_pk_rec* pk);
// makes sure `pk` can prove `batch` proofs in lock-step: batch - 1 member workspaces and the shared buffers (allocated on
// first use, kept with the key); caller holds the context lock, device bound
int pk_ensure_batch(zk_ctx* c, zk_pk_rec* pk, uint32_t batch);
// transcript_repr of a key made or read here: halo2.s own hash of the pinned verifying key (vkrepr.h); a stand-in for the shapes
// that rendering does not cover; a host-supplied value replaces either
Fr pk_standin_transcript_repr(const zk_pk_rec* pk);
