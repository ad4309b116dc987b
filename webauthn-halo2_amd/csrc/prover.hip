// prover.hip — keygen and create_proof on the device-resident engine.
//
// Host orchestration (C++) of the kernels in msm.hip / ntt.hip / quotient.hip /
// prover_kernels.hip / poly.hip, mirroring halo2_proofs `plonk::keygen_vk`,
// `keygen_pk` and `plonk::create_proof` as the reference calls them:
//   keygen        halo2-circuits/src/ecc/ecdsa_p256.rs:259-260
//   create_proof  halo2-circuits/src/ecc/ecdsa_p256.rs:366-373 (EvmTranscript + ProverGWC)
//                 halo2-circuits/src/ecc/ecdsa_p256.rs:416-423 (Blake2bWrite + ProverSHPLONK)
// Phase structure, transcript order and RNG draw order: SURVEY.md §3.3 / App. A.3
// (the test oracle restates the same flow in Python; proofs are byte-compared against it).
// Polynomials never leave HBM: the host sees commitments (64 B), evaluations
// (32 B) and challenges only.  Randomness is a ChaCha20 stream (rand_chacha's
// ChaCha20Rng layout): one 64-byte block per Fr::random, drawn on the host for
// the handful of blinding rows and on the device for the n-coefficient random
// polynomial — the engine's kernels themselves consume no randomness.
#include <stdlib.h>

#include <algorithm>
#include <deque>
#include <functional>
#include <memory>
#include <unordered_map>
#include <vector>

#include "pk.h"
#include "transcript.h"
#include "vkrepr.h"

using namespace zk;

#ifdef ZK_HOST_TRACE  // host-side stage stamps of the proof's last phases (experiments only: tools/ab_variants.sh ... "-DZK_HOST_TRACE")
#include <chrono>
#include <cstdio>
static std::chrono::steady_clock::time_point& ht_last() {
    static thread_local std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    return t;
}
#define HT(name)                                                                                                  \
    do {                                                                                                          \
        const auto ht_now = std::chrono::steady_clock::now();                                                     \
        fprintf(stderr, "HT %-24s +%7.1f us\n", name, std::chrono::duration<double, std::micro>(ht_now - ht_last()).count()); \
        ht_last() = ht_now;                                                                                       \
    } while (0)
#else
#define HT(name) do { } while (0)
#endif

namespace {

// ---------------------------------------------------------------- kernels ---
__global__ void sigma_kernel(const uint2* __restrict__ map, const Fr* __restrict__ tw, const Fr* __restrict__ dpow,
                             Fr* __restrict__ out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint2 m = map[i];
    fe_store(out + i, fe_mul(fe_load(dpow + m.x), fe_load(tw + m.y)));
}

// ------------------------------------------------------------ small utils ---
bool commit(zk_ctx* c, const Fr* poly, size_t len, int basis, G1Affine* out) {
    G1Jac j;
    if (ctx_msm_device(c, poly, basis == ZK_BASIS_LAGRANGE ? c->g_lagrange : c->g, len, &j) != ZK_OK) return false;
    *out = g1_jac_to_affine_host(j);
    return true;
}

Fr fr_pow(Fr a, uint64_t e) { return fe_pow_u64(a, e); }

Fr fr_delta() {  // 7^(2^28): generator of the odd-order subgroup
    static const Fr delta = [] {
        Fr d = fr_from_u64(7);
        for (int i = 0; i < 28; i++) d = fe_sqr(d);
        return d;
    }();
    return delta;
}

void fr_to_le_bytes(const Fr& mont, uint8_t out[32]) {
    const Fr c = fe_from_mont(mont);
    memcpy(out, c.v, 32);
}

// Integer order of canonical values (halo2curves Fr: Ord)
bool fr_less(const Fr& a_mont, const Fr& b_mont) {
    const Fr a = fe_from_mont(a_mont), b = fe_from_mont(b_mont);
    for (int i = 7; i >= 0; i--) {
        if (a.v[i] != b.v[i]) return a.v[i] < b.v[i];
    }
    return false;
}

// The Lagrange basis over `pts` as coefficient vectors: basis[j] = L_j(X) = prod_{i != j} (X - pts_i) / (pts_j - pts_i).  A
// rotation set's commitments share their points, so this is paid once per set — with ONE field inversion for all the
// denominators — and a commitment's interpolant is sum_j eval_j L_j (m^2 products): the same field elements
// halo2's lagrange_interpolate computes per commitment, hence the same bytes.
std::vector<std::vector<Fr>> lagrange_basis(const std::vector<Fr>& pts) {
    const size_t m = pts.size();
    std::vector<std::vector<Fr>> basis(m);
    std::vector<Fr> den(m, Fr::one()), pre(m);
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
    // all 1 / den_j from one inversion (distinct points: no denominator is zero)
    Fr run = Fr::one();
    for (size_t j = 0; j < m; j++) {
        pre[j] = run;
        run = fe_mul(run, den[j]);
    }
    Fr inv = fe_inv_fast(run);
    for (size_t j = m; j-- > 0;) {
        const Fr dj = fe_mul(inv, pre[j]);
        inv = fe_mul(inv, den[j]);
        for (Fr& cf : basis[j]) cf = fe_mul(cf, dj);
    }
    return basis;
}

// affine forms of `cnt` Jacobian points with ONE field inversion for the whole batch (Montgomery's trick over the z's)
void jac_batch_to_affine(const G1Jac* js, uint32_t cnt, G1Affine* af) {
    Fq pre[MSM_MAX_BATCH];
    Fq run = Fq::one();
    for (uint32_t q = 0; q < cnt; q++) {
        pre[q] = run;
        if (!js[q].z.is_zero()) run = fe_mul(run, js[q].z);
    }
    Fq inv = fe_inv_fast(run);
    for (uint32_t q = cnt; q-- > 0;) {
        if (js[q].z.is_zero()) {
            af[q].x = Fq::zero();
            af[q].y = Fq::zero();
            continue;
        }
        const Fq zi = fe_mul(inv, pre[q]);
        inv = fe_mul(inv, js[q].z);
        const Fq zi2 = fe_sqr(zi);
        af[q].x = fe_mul(js[q].x, zi2);
        af[q].y = fe_mul(js[q].y, fe_mul(zi2, zi));
    }
}

Fr eval_small(const std::vector<Fr>& c, const Fr& x) {
    Fr acc = Fr::zero();
    for (size_t i = c.size(); i-- > 0;) acc = fe_add(fe_mul(acc, x), c[i]);
    return acc;
}

Fr vanishing_eval(const std::vector<Fr>& pts, const Fr& x) {
    Fr acc = Fr::one();
    for (const Fr& p : pts) acc = fe_mul(acc, fe_sub(x, p));
    return acc;
}

}  // namespace

static void bb_destroy(BatchBufs* bb) {
    if (!bb) return;
    if (bb->lk_u32) hipFree(bb->lk_u32);
    if (bb->d_gp_items) hipFree(bb->d_gp_items);
    if (bb->gp_scal) hipFree(bb->gp_scal);
    if (bb->gp_host) hipHostFree(bb->gp_host);
    if (bb->d_evargs) hipFree(bb->d_evargs);
    if (bb->h_evargs) hipHostFree(bb->h_evargs);
    if (bb->ev_scratch) hipFree(bb->ev_scratch);
    if (bb->ev_out) hipFree(bb->ev_out);
    if (bb->tail_host) hipHostFree(bb->tail_host);
    delete bb;
}

void pk_destroy(zk_pk_rec* pk) {
    if (!pk) return;
    for (zk_pk_rec* m : pk->members) pk_destroy(m);  // (a member's key half aliases this record's: only its workspace goes)
    pk->members.clear();
    bb_destroy(pk->bb);
    pk->bb = nullptr;
    for (Fr* p : pk->dev) hipFree(p);
    if (pk->tail_host) hipHostFree(pk->tail_host);
    if (pk->rows_host) hipHostFree(pk->rows_host);
    if (pk->rows_dev) hipFree(pk->rows_dev);
    if (pk->lk_u32) hipFree(pk->lk_u32);
    if (pk->gp_host) hipHostFree(pk->gp_host);
    if (pk->d_gp_items) hipFree(pk->d_gp_items);
    if (pk->d_qargs) hipFree(pk->d_qargs);
    if (pk->d_batch_args) hipFree(pk->d_batch_args);
    if (pk->h_batch_args) hipHostFree(pk->h_batch_args);
    if (pk->h_qargs) hipHostFree(pk->h_qargs);
    if (pk->d_lc_terms) hipFree(pk->d_lc_terms);
    if (pk->h_lc_terms) hipHostFree(pk->h_lc_terms);
    if (pk->d_evargs) hipFree(pk->d_evargs);
    if (pk->h_evargs) hipHostFree(pk->h_evargs);
    delete pk;
}

void pk_destroy_all(zk_ctx* c) {
    for (auto& kv : c->pks) pk_destroy(kv.second);
    c->pks.clear();
}

// halo2's transcript_repr of a key made (or read) here: the hash of the pinned verifying key's Debug rendering
// (vkrepr.h) — every shape, never-enabled gate columns included (round 4: their combined selectors are rendered as
// compress_selectors builds them; the stand-in hash of earlier rounds is gone).  A host-supplied value still replaces it.
Fr pk_standin_transcript_repr(const zk_pk_rec* pk) { return vkrepr::transcript_repr(pk->lay, pk->fixed_commit, pk->perm_commit); }

int pk_alloc_workspace(zk_ctx* c, zk_pk_rec* pk) {
    const Layout& lay = pk->lay;
    const uint32_t n = lay.n, N = 4 * n, T = 1u << lay.lookup_bits;
    Dev d{c, pk};
    auto fail = [&](int code) { return code; };  // the caller destroys the key
    // ---- prover workspace
    for (uint32_t j = 0; j < lay.n_adv; j++) {
        pk->adv_val.push_back(d.alloc(n));
        pk->adv_poly.push_back(d.alloc(n));
        pk->adv_coset.push_back(d.alloc(N));
    }
    for (uint32_t ci = 0; ci < lay.n_chunks; ci++) {
        pk->z_val.push_back(d.alloc(n));
        pk->z_poly.push_back(d.alloc(n));
        pk->z_coset.push_back(d.alloc(N));
    }
    for (uint32_t l = 0; l < lay.n_lookups; l++) {
        pk->lk_in.push_back(lay.single ? d.alloc(n) : nullptr);
        pk->lk_ap.push_back(d.alloc(n));
        pk->lk_ap_poly.push_back(d.alloc(n));
        pk->lk_ap_coset.push_back(d.alloc(N));
        pk->lk_sp.push_back(d.alloc(n));
        pk->lk_sp_poly.push_back(d.alloc(n));
        pk->lk_sp_coset.push_back(d.alloc(N));
        pk->lk_z.push_back(d.alloc(n));
        pk->lk_z_poly.push_back(d.alloc(n));
        pk->lk_z_coset.push_back(d.alloc(N));
    }
    pk->random_poly = d.alloc(n);
    pk->h_ext = d.alloc(N);
    pk->h_comb = d.alloc(n);
    pk->t_num = d.alloc(n);
    pk->t_den = d.alloc(n);
    pk->t_frac = d.alloc(n);
    pk->t_a = d.alloc(n);
    pk->t_b = d.alloc(n);
    pk->t_small = d.alloc(n / 16 + 8192);
    pk->kd_scratch = d.alloc((size_t)KD_MAX_BATCH * kate_division_scratch(n));
    {
        const uint32_t nprod = lay.n_chunks + lay.n_lookups;
        for (uint32_t p = 0; p < nprod; p++) {
            pk->gp_num.push_back(d.alloc(n));
            pk->gp_den.push_back(d.alloc(n));
            pk->gp_loc_p.push_back(d.alloc(n));
            pk->gp_loc_r.push_back(d.alloc(n));
        }
        pk->gp_tot = d.alloc((size_t)2 * gp_blocks(n) * nprod);
        pk->gp_scal = d.alloc((size_t)4 * nprod);
        if (hipHostMalloc(&pk->gp_host, (size_t)2 * nprod * sizeof(Fr)) != hipSuccess ||
            hipMalloc(&pk->d_gp_items, nprod * sizeof(GpItem)) != hipSuccess)
            return fail(ZK_ENOMEM);
    }
    if (d.rc) return fail(d.rc);
    if (hipHostMalloc(&pk->tail_host, (pk->max_evals + 16) * sizeof(Fr)) != hipSuccess) return fail(ZK_ENOMEM);
    if (hipHostMalloc(&pk->rows_host, (size_t)ROWS_BLOCKS * ROWS_CAP * sizeof(RowEntry)) != hipSuccess ||
        hipMalloc(&pk->rows_dev, (size_t)ROWS_BLOCKS * ROWS_CAP * sizeof(RowEntry)) != hipSuccess)
        return fail(ZK_ENOMEM);
    if (hipHostMalloc(&pk->h_evargs, pk->max_evals * sizeof(EvalItem)) != hipSuccess ||
        hipMalloc(&pk->d_evargs, pk->max_evals * sizeof(EvalItem)) != hipSuccess)
        return fail(ZK_ENOMEM);
    pk->lc_cap = 2 * (pk->max_evals + 8);
    if (hipHostMalloc(&pk->h_lc_terms, pk->lc_cap * sizeof(LcTerm)) != hipSuccess ||
        hipMalloc(&pk->d_lc_terms, pk->lc_cap * sizeof(LcTerm)) != hipSuccess)
        return fail(ZK_ENOMEM);
    pk->ev_scratch = d.alloc((size_t)pk->max_evals * eval_blocks(n));
    pk->ev_out = d.alloc(pk->max_evals);
    if (d.rc) return fail(d.rc);
    {
        // per lookup: six arrays of T + 2 words and 3 x blocks block sums; one error flag for all
        const uint32_t stride = 6 * (T + 2) + 3 * (T / 1024 + 2);
        if (hipMalloc(&pk->lk_u32, ((size_t)stride * lay.n_lookups + 4) * 4) != hipSuccess) return fail(ZK_ENOMEM);
        uint32_t* b = pk->lk_u32;
        pk->lks.hist = b;
        pk->lks.present = b + (T + 2);
        pk->lks.absent = b + 2 * (T + 2);
        pk->lks.off = b + 3 * (T + 2);
        pk->lks.dex = b + 4 * (T + 2);
        pk->lks.aex = b + 5 * (T + 2);
        pk->lks.bsum = b + 6 * (T + 2);
        pk->lks.stride = stride;
        pk->lks.err = b + (size_t)stride * lay.n_lookups;
    }
    if (hipMalloc(&pk->d_qargs, sizeof(QuotientArgs)) != hipSuccess || hipHostMalloc(&pk->h_qargs, sizeof(QuotientArgs)) != hipSuccess)
        return fail(ZK_ENOMEM);
    {
        size_t bytes = (size_t)lay.n_chunks * sizeof(PermArgs);
        bytes = std::max(bytes, (size_t)lay.n_lookups * sizeof(LkNumDenArgs));
        bytes = std::max(bytes, (size_t)lay.n_adv * sizeof(CopyPair));
        pk->batch_args_bytes = bytes;
        if (hipMalloc(&pk->d_batch_args, bytes) != hipSuccess || hipHostMalloc(&pk->h_batch_args, bytes) != hipSuccess) return fail(ZK_ENOMEM);
    }
    return ZK_OK;
}

// a further workspace for the same key: the record is copied (the key half stays shared — nothing of it is in the copy's
// `dev` list), every workspace member is reset and allocated afresh
static zk_pk_rec* pk_make_member(zk_ctx* c, const zk_pk_rec* pk) {
    zk_pk_rec* m = new (std::nothrow) zk_pk_rec(*pk);
    if (!m) return nullptr;
    m->is_member = true;
    m->dev.clear();
    m->members.clear();
    m->bb = nullptr;
    for (auto* v : {&m->adv_val, &m->adv_poly, &m->adv_coset, &m->z_val, &m->z_poly, &m->z_coset, &m->lk_in, &m->lk_ap, &m->lk_ap_poly,
                    &m->lk_ap_coset, &m->lk_sp, &m->lk_sp_poly, &m->lk_sp_coset, &m->lk_z, &m->lk_z_poly, &m->lk_z_coset, &m->lk_in_coset,
                    &m->gp_num, &m->gp_den, &m->gp_loc_p, &m->gp_loc_r})
        v->clear();
    m->random_poly = m->h_ext = m->h_comb = m->t_num = m->t_den = m->t_frac = m->t_a = m->t_b = m->t_small = m->kd_scratch = nullptr;
    m->tail_host = nullptr;
    m->rows_host = m->rows_dev = nullptr;
    m->lk_u32 = nullptr;
    m->gp_tot = m->gp_scal = m->gp_host = nullptr;
    m->d_gp_items = nullptr;
    m->h_batch_args = m->d_batch_args = nullptr;
    m->d_qargs = m->h_qargs = nullptr;
    m->d_evargs = m->h_evargs = nullptr;
    m->d_lc_terms = m->h_lc_terms = nullptr;
    m->ev_scratch = m->ev_out = nullptr;
    m->lc_used = 0;
    if (pk_alloc_workspace(c, m) != ZK_OK) {
        pk_destroy(m);
        return nullptr;
    }
    return m;
}

int pk_ensure_batch(zk_ctx* c, zk_pk_rec* pk, uint32_t batch) {
    if (batch <= 1) return ZK_OK;
    const Layout& lay = pk->lay;
    const uint32_t n = lay.n, T = 1u << lay.lookup_bits, nprod = lay.n_chunks + lay.n_lookups;
    if ((pk->members.size() + 1 < batch || !pk->bb || pk->bb->cap < batch) && !c->poly_spare.empty()) ctx_release_spares(c);
    while (pk->members.size() + 1 < batch) {
        zk_pk_rec* m = pk_make_member(c, pk);
        if (!m) return ZK_ENOMEM;
        pk->members.push_back(m);
    }
    if (pk->bb && pk->bb->cap >= batch) return ZK_OK;
    aud_sync(c, c->stream);
    bb_destroy(pk->bb);
    pk->bb = nullptr;
    BatchBufs* bb = new (std::nothrow) BatchBufs();
    if (!bb) return ZK_ENOMEM;
    pk->bb = bb;  // (freed with the key whatever happens below)
    const size_t nl = (size_t)batch * lay.n_lookups, np = (size_t)batch * nprod, ne = (size_t)batch * pk->max_evals;
    const uint32_t stride = 6 * (T + 2) + 3 * (T / 1024 + 2);
    if (hipMalloc(&bb->lk_u32, ((size_t)stride * nl + 4) * 4) != hipSuccess || hipMalloc(&bb->d_gp_items, np * sizeof(GpItem)) != hipSuccess ||
        hipMalloc(&bb->gp_scal, 4 * np * sizeof(Fr)) != hipSuccess || hipHostMalloc(&bb->gp_host, 2 * np * sizeof(Fr)) != hipSuccess ||
        hipMalloc(&bb->d_evargs, ne * sizeof(EvalItem)) != hipSuccess || hipHostMalloc(&bb->h_evargs, ne * sizeof(EvalItem)) != hipSuccess ||
        hipMalloc(&bb->ev_scratch, ne * eval_blocks(n) * sizeof(Fr)) != hipSuccess || hipMalloc(&bb->ev_out, ne * sizeof(Fr)) != hipSuccess ||
        hipHostMalloc(&bb->tail_host, ne * sizeof(Fr)) != hipSuccess)
        return ZK_ENOMEM;
    uint32_t* b = bb->lk_u32;
    bb->lks.hist = b;
    bb->lks.present = b + (T + 2);
    bb->lks.absent = b + 2 * (T + 2);
    bb->lks.off = b + 3 * (T + 2);
    bb->lks.dex = b + 4 * (T + 2);
    bb->lks.aex = b + 5 * (T + 2);
    bb->lks.bsum = b + 6 * (T + 2);
    bb->lks.stride = stride;
    bb->lks.err = b + (size_t)stride * nl;
    bb->cap = batch;
    return ZK_OK;
}

// =================================================================== keygen ==

ZK_API(zk_keygen, (zk_ctx* c, const zk_circuit_params* params, const uint64_t* fixed_canonical, size_t n_fixed_columns, const uint32_t* copies, size_t n_copies, zk_pk* out), (c, params, fixed_canonical, n_fixed_columns, copies, n_copies, out)) {
    if (!c || !params || !fixed_canonical || !out || (n_copies && !copies)) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    int rc = ctx_bind(c);
    if (rc) return rc;
    ctx_release_spares(c);  // parked vectors are reclaimable: give them back before the key and its workspace are allocated
    Layout lay;
    if (params->num_advice > 1 && 2 * (uint64_t)params->num_idle_gate_columns > params->num_advice) return ZK_ELAYOUT;  // zkmi355.h: more never-enabled selectors than used ones
    if (!lay.init(*params)) return ZK_EINVAL;
    if (n_fixed_columns != lay.n_fix) return ZK_EINVAL;  // fixed_canonical holds n_fixed_columns x n x 4 limbs
    if (c->srs_k != (int)lay.k) return ZK_ESTATE;
    const uint32_t n = lay.n, N = 4 * n, T = 1u << lay.lookup_bits;
    // the lookup path is specialised to halo2-lib's range table: 0..T-1 then zeros
    {
        const uint64_t* tab = fixed_canonical + (size_t)lay.fx_table * n * 4;
        for (uint32_t r = 0; r < n; r++) {
            const uint64_t want = r < T ? r : 0;
            if (tab[4 * r] != want || tab[4 * r + 1] || tab[4 * r + 2] || tab[4 * r + 3]) return ZK_EINVAL;
        }
    }
    // the gate selectors must be what the key's closed-form layout assumes of them (pk.h layout_selectors_fit): 0 / 1 columns
    // that halo2's compress_selectors would leave one fixed column each
    if (!lay.single) {
        std::vector<std::vector<uint8_t>> bits(lay.A, std::vector<uint8_t>(n / 8, 0));
        for (uint32_t j = 0; j < lay.A; j++) {
            if (lay.fx_sel[j] == NO_SELECTOR) continue;
            const uint64_t* col = fixed_canonical + (size_t)lay.fx_sel[j] * n * 4;
            for (uint32_t r = 0; r < n; r++) {
                if (col[4 * r] > 1 || col[4 * r + 1] || col[4 * r + 2] || col[4 * r + 3]) return ZK_EINVAL;  // not a selector column
                if (col[4 * r]) bits[j][r >> 3] |= (uint8_t)(1u << (r & 7));
            }
        }
        if (!layout_selectors_fit(lay, bits)) return ZK_ELAYOUT;
    }
    const uint32_t m = (uint32_t)lay.perm_cols.size();
    for (size_t i = 0; i < n_copies; i++) {
        const uint32_t* e = copies + 4 * i;
        if (e[0] >= m || e[2] >= m || e[1] >= lay.usable || e[3] >= lay.usable) return ZK_EINVAL;
    }
    zk_pk_rec* pk = new (std::nothrow) zk_pk_rec();
    if (!pk) return ZK_ENOMEM;
    pk->lay = lay;
    pk->srs_gen = c->srs_gen;
    pk->max_evals = (uint32_t)(lay.advice_queries.size() + lay.n_fix + lay.perm_cols.size() + 3 * lay.n_chunks +
                               5 * lay.n_lookups + 16);
    Dev d{c, pk};
    hipStream_t st = c->stream;
    const Fr* tw = nullptr;
    const Fr* tw_ext = nullptr;
    if ((rc = ctx_get_twiddles(c, lay.k, &tw)) || (rc = ctx_get_twiddles(c, lay.ext_k, &tw_ext))) {
        pk_destroy(pk);
        return rc;
    }
    auto fail = [&](int code) {
        aud_sync(c, st);
        pk_destroy(pk);
        return code;
    };

    // ---- fixed columns: values -> commitment, coefficients, extended coset
    for (uint32_t f = 0; f < lay.n_fix; f++) {
        Fr *v = d.alloc(n), *p = d.alloc(n), *e = d.alloc(N);
        if (d.rc) return fail(d.rc);
        pk->fixed_val.push_back(v);
        pk->fixed_poly.push_back(p);
        pk->fixed_coset.push_back(e);
        hipMemcpyAsync(v, fixed_canonical + (size_t)f * n * 4, (size_t)n * sizeof(Fr), hipMemcpyHostToDevice, st);
        launch_to_mont(v, n, st);
    }
    // ---- permutation: halo2 permutation::keygen::Assembly (cycle merging), then sigma = delta^c' w^r'
    {
        std::vector<uint2> mapping((size_t)m * n), aux((size_t)m * n);
        std::vector<uint32_t> sizes((size_t)m * n, 1);
        for (uint32_t col = 0; col < m; col++)
            for (uint32_t r = 0; r < n; r++) mapping[(size_t)col * n + r] = aux[(size_t)col * n + r] = make_uint2(col, r);
        auto at = [&](uint2 p) { return (size_t)p.x * n + p.y; };
        auto same = [](uint2 a, uint2 b) { return a.x == b.x && a.y == b.y; };
        for (size_t i = 0; i < n_copies; i++) {
            uint2 l = make_uint2(copies[4 * i], copies[4 * i + 1]), r = make_uint2(copies[4 * i + 2], copies[4 * i + 3]);
            uint2 lc = aux[at(l)], rc2 = aux[at(r)];
            if (same(lc, rc2)) continue;
            if (sizes[at(lc)] < sizes[at(rc2)]) {
                std::swap(lc, rc2);
                std::swap(l, r);
            }
            sizes[at(lc)] += sizes[at(rc2)];
            uint2 it = rc2;
            for (;;) {
                aux[at(it)] = lc;
                it = mapping[at(it)];
                if (same(it, rc2)) break;
            }
            std::swap(mapping[at(l)], mapping[at(r)]);
        }
        std::vector<Fr> dpow(m);
        Fr dl = Fr::one();
        const Fr delta = fr_delta();
        for (uint32_t col = 0; col < m; col++) {
            dpow[col] = dl;
            dl = fe_mul(dl, delta);
        }
        uint2* d_map = nullptr;
        Fr* d_dpow = d.alloc(m);
        if (d.rc || hipMalloc(&d_map, (size_t)n * sizeof(uint2)) != hipSuccess) return fail(ZK_ENOMEM);
        hipMemcpyAsync(d_dpow, dpow.data(), m * sizeof(Fr), hipMemcpyHostToDevice, st);
        for (uint32_t col = 0; col < m; col++) {
            Fr *v = d.alloc(n), *p = d.alloc(n), *e = d.alloc(N);
            if (d.rc) {
                hipFree(d_map);
                return fail(d.rc);
            }
            pk->sigma_val.push_back(v);
            pk->sigma_poly.push_back(p);
            pk->sigma_coset.push_back(e);
            hipMemcpyAsync(d_map, &mapping[(size_t)col * n], (size_t)n * sizeof(uint2), hipMemcpyHostToDevice, st);
            hipLaunchKernelGGL(sigma_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_map, tw, d_dpow, v, n);
            aud_sync(c, st);  // d_map is reused
        }
        hipFree(d_map);
    }
    // ---- commitments (vk) and polynomial forms (pk)
    auto finish_col = [&](Fr* v, Fr* p, Fr* e, G1Affine* cm) -> int {
        if (!commit(c, v, n, ZK_BASIS_LAGRANGE, cm)) return ZK_EHIP;
        hipMemcpyAsync(p, v, (size_t)n * sizeof(Fr), hipMemcpyDeviceToDevice, st);
        int r2 = ctx_ntt(c, p, n, p, lay.k, true, false, n);
        if (r2) return r2;
        return ctx_ntt(c, p, n, e, lay.ext_k, false, true, N);
    };
    pk->fixed_commit.resize(lay.n_fix);
    pk->perm_commit.resize(m);
    for (uint32_t f = 0; f < lay.n_fix; f++)
        if ((rc = finish_col(pk->fixed_val[f], pk->fixed_poly[f], pk->fixed_coset[f], &pk->fixed_commit[f]))) return fail(rc);
    for (uint32_t col = 0; col < m; col++)
        if ((rc = finish_col(pk->sigma_val[col], pk->sigma_poly[col], pk->sigma_coset[col], &pk->perm_commit[col]))) return fail(rc);
    // ---- l_0, l_last, l_active (= 1 - l_last - l_blind) cosets
    {
        std::vector<Fr> tmp(n, Fr::zero());
        Fr* scratch_n = d.alloc(n);
        pk->l0_coset = d.alloc(N);
        pk->l_last_coset = d.alloc(N);
        pk->l_active_coset = d.alloc(N);
        if (d.rc) return fail(d.rc);
        auto make = [&](Fr* dst) -> int {
            hipMemcpyAsync(scratch_n, tmp.data(), (size_t)n * sizeof(Fr), hipMemcpyHostToDevice, st);
            aud_sync(c, st);
            int r2 = ctx_ntt(c, scratch_n, n, scratch_n, lay.k, true, false, n);
            if (r2) return r2;
            return ctx_ntt(c, scratch_n, n, dst, lay.ext_k, false, true, N);
        };
        tmp[0] = Fr::one();
        if ((rc = make(pk->l0_coset))) return fail(rc);
        tmp[0] = Fr::zero();
        tmp[lay.usable] = Fr::one();  // row n - (bf + 1)
        if ((rc = make(pk->l_last_coset))) return fail(rc);
        for (uint32_t r = 0; r < n; r++) tmp[r] = r < lay.usable ? Fr::one() : Fr::zero();
        if ((rc = make(pk->l_active_coset))) return fail(rc);
    }
    pk->transcript_repr = pk_standin_transcript_repr(pk);
    if ((rc = pk_alloc_workspace(c, pk))) return fail(rc);
    if (aud_sync(c, st) != hipSuccess || hipGetLastError() != hipSuccess) return fail(ZK_EHIP);
    const uint64_t h = c->next_handle++;
    c->pks[h] = pk;
    *out = h;
    return ZK_OK;
}

ZK_API(zk_pk_free, (zk_ctx* c, zk_pk h), (c, h)) {
    if (!c) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    auto it = c->pks.find(h);
    if (it == c->pks.end()) return ZK_EINVAL;
    ctx_bind(c);
    aud_sync(c, c->stream);
    pk_destroy(it->second);
    c->pks.erase(it);
    return ZK_OK;
}

ZK_API(zk_vk_export, (zk_ctx* c, zk_pk h, uint64_t* fixed_commitments, uint64_t* perm_commitments, uint64_t transcript_repr[4], uint32_t counts[2]), (c, h, fixed_commitments, perm_commitments, transcript_repr, counts)) {
    if (!c) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    auto it = c->pks.find(h);
    if (it == c->pks.end()) return ZK_EINVAL;
    zk_pk_rec* pk = it->second;
    if (pk->srs_gen != c->srs_gen) return ZK_ESTATE;  // the SRS was replaced after this key was made
    if (counts) {
        counts[0] = (uint32_t)pk->fixed_commit.size();
        counts[1] = (uint32_t)pk->perm_commit.size();
    }
    if (fixed_commitments) memcpy(fixed_commitments, pk->fixed_commit.data(), pk->fixed_commit.size() * sizeof(G1Affine));
    if (perm_commitments) memcpy(perm_commitments, pk->perm_commit.data(), pk->perm_commit.size() * sizeof(G1Affine));
    if (transcript_repr) memcpy(transcript_repr, &pk->transcript_repr, 32);
    return ZK_OK;
}

int pk_ensure_cosets3(zk_ctx* c, zk_pk_rec* pk) {
    const Layout& lay = pk->lay;
    if (lay.n_h != 3) return ZK_EINVAL;
    auto to_members = [&]() {  // by-value copies of the record (pk_make_member): the key half is written through
        for (zk_pk_rec* m : pk->members) {
            m->fixed_c3 = pk->fixed_c3;
            m->sigma_c3 = pk->sigma_c3;
            m->l0_c3 = pk->l0_c3;
            m->l_last_c3 = pk->l_last_c3;
            m->l_active_c3 = pk->l_active_c3;
        }
    };
    if (pk->fixed_c3.size() == lay.n_fix && pk->l_active_c3) {
        to_members();
        return ZK_OK;
    }
    const size_t n = lay.n;
    Dev d{c, pk};
    std::vector<const Fr*> src;
    std::vector<Fr*> dst;
    auto add = [&](const Fr* s) {
        Fr* t = d.alloc(3 * n);
        src.push_back(s);
        dst.push_back(t);
        return t;
    };
    std::vector<Fr*> fx, sg;
    for (uint32_t f = 0; f < lay.n_fix; f++) fx.push_back(add(pk->fixed_coset[f]));
    for (size_t p = 0; p < lay.perm_cols.size(); p++) sg.push_back(add(pk->sigma_coset[p]));
    Fr *a0 = add(pk->l0_coset), *a1 = add(pk->l_last_coset), *a2 = add(pk->l_active_coset);
    if (d.rc) return d.rc;  // (what was allocated stays on the key's list and is freed with it)
    launch_coset3_relayout(src.data(), dst.data(), (uint32_t)src.size(), (uint32_t)n, c->stream);
    if (c->audit.on) {
        std::vector<const void*> rd(src.begin(), src.end()), wr(dst.begin(), dst.end());
        c->audit.op_v(c->stream, rd.data(), rd.size(), wr.data(), wr.size(), "key cosets -> coset-major");
    }
    pk->fixed_c3 = fx;
    pk->sigma_c3 = sg;
    pk->l0_c3 = a0;
    pk->l_last_c3 = a1;
    pk->l_active_c3 = a2;
    to_members();
    return ZK_OK;
}

// Evaluator::evaluate_h (+ divide_by_vanishing_poly when `divide`) over resident extended cosets: the key's fixed /
// sigma / l_* cosets and the caller's advice, permutation-product and lookup cosets.  Enqueued on the context stream.
int pk_quotient(zk_ctx* c, zk_pk_rec* pk, const QuotientCosets& qc, const Fr& beta, const Fr& gamma, const Fr& y, bool divide, Fr* out) {
    const Layout& lay = pk->lay;
    if (qc.adv.size() != lay.n_adv || qc.z.size() != lay.n_chunks || qc.lk_a.size() != lay.n_lookups ||
        qc.lk_s.size() != lay.n_lookups || qc.lk_z.size() != lay.n_lookups)
        return ZK_EINVAL;
    const Fr* xs = nullptr;
    int rc = ctx_get_coset_points(c, lay.ext_k, &xs);
    if (rc) return rc;
    QuotientArgs& q = *pk->h_qargs;  // pinned: the upload below does not stall the host (the previous use is complete)
    memset(&q, 0, sizeof(q) - sizeof(q.ypow));
    q.log_ext = lay.ext_k;
    q.n_gate = lay.n_gate;
    q.n_adv = lay.n_adv;
    q.n_chunks = lay.n_chunks;
    q.chunk_len = lay.chunk_len;
    q.n_perm = (uint32_t)lay.perm_cols.size();
    q.n_lookups = lay.n_lookups;
    q.single = lay.single ? 1 : 0;
    q.last_rot = lay.last_rot;
    q.fx_table = lay.fx_table;
    q.fx_qlookup = lay.fx_qlookup;
    for (uint32_t j = 0; j < lay.n_adv; j++) q.adv[j] = qc.adv[j];
    const bool c3 = qc.cosets3;
    if (c3 && (lay.n_h != 3 || pk->fixed_c3.size() != lay.n_fix)) return ZK_EINVAL;  // (pk_ensure_cosets3 first)
    for (uint32_t f = 0; f < lay.n_fix; f++) q.fix[f] = c3 ? pk->fixed_c3[f] : pk->fixed_coset[f];
    for (uint32_t j = 0; j < lay.n_gate; j++) q.fx_sel[j] = lay.gate_sel[j];
    for (uint32_t p = 0; p < q.n_perm; p++) {
        q.sigma[p] = c3 ? pk->sigma_c3[p] : pk->sigma_coset[p];
        const Col& col = lay.perm_cols[p];
        q.perm_val[p] = col.fixed ? q.fix[col.idx] : qc.adv[col.idx];
    }
    for (uint32_t ci = 0; ci < lay.n_chunks; ci++) q.z[ci] = qc.z[ci];
    for (uint32_t l = 0; l < lay.n_lookups; l++) {
        q.lk_z[l] = qc.lk_z[l];
        q.lk_a[l] = qc.lk_a[l];
        q.lk_s[l] = qc.lk_s[l];
        q.lk_in[l] = lay.single ? nullptr : qc.adv[lay.n_gate + l];
    }
    q.l0 = c3 ? pk->l0_c3 : pk->l0_coset;
    q.l_last = c3 ? pk->l_last_c3 : pk->l_last_coset;
    q.l_active = c3 ? pk->l_active_c3 : pk->l_active_coset;
    q.xs = xs;
    // the kernel works in the carry-free field's internal form (x * 2^261): its constants are handed over times 32
    const Fr k32 = fr_from_u64(32);
    q.beta = fe_mul(beta, k32);
    q.gamma = fe_mul(gamma, k32);
    q.delta = fe_mul(fr_delta(), k32);
    // 1 / ((zeta w_ext^i)^n - 1): zeta^n * (w_ext^n)^i, w_ext^n is a primitive 4th root
    if (!pk->t_inv_ready) {  // constants of the key's domain: made once, not once per proof
        const Fr zn = fe_pow_u64(c->zeta, lay.n);
        const Fr w4 = fe_pow_u64(fr_omega(lay.ext_k), lay.n);
        Fr cur = zn;
        for (int i = 0; i < 4; i++) {
            pk->t_inv[i] = fe_inv_fast(fe_sub(cur, Fr::one()));
            cur = fe_mul(cur, w4);
        }
        pk->t_inv_ready = true;
    }
    // standard form: the product by it also converts the row back (quotient.hip); 1 = no division
    for (int i = 0; i < 4; i++) q.t_inv[i] = divide ? pk->t_inv[i] : Fr::one();
    q.divide = divide ? 1 : 0;
    q.n_terms = quotient_terms(lay.n_gate, lay.n_chunks, lay.n_lookups);
    if (q.n_terms > MAX_TERMS) return ZK_EINVAL;
    Fr yp = k32;
    for (uint32_t j = q.n_terms; j-- > 0;) {  // ypow[j] = 32 y^(T - 1 - j)
        q.ypow[j] = yp;
        yp = fe_mul(yp, y);
    }
    q.out = out;
    const uint32_t log_slices = quotient_log_slices(lay.ext_k, lay.n_gate);
    if (log_slices) {
        const Fr dstep = fe_pow_u64(fr_delta(), lay.chunk_len);
        Fr dc = k32;
        for (uint32_t ci = 0; ci < lay.n_chunks; ci++) {
            q.delta_chunk[ci] = dc;
            dc = fe_mul(dc, dstep);
        }
    }
    hipStream_t st = c->stream;
    hipEventRecord(c->ev[ZK_T_QUOTIENT][0], st);
    const size_t bytes = sizeof(q) - sizeof(q.ypow) + (size_t)q.n_terms * sizeof(Fr);
    if (hipMemcpyAsync(pk->d_qargs, &q, bytes, hipMemcpyHostToDevice, st) != hipSuccess) return ZK_EHIP;
    launch_quotient_dev(pk->d_qargs, lay.ext_k, log_slices, st, c3);
    hipEventRecord(c->ev[ZK_T_QUOTIENT][1], st);
    c->ev_valid[ZK_T_QUOTIENT] = true;
    return ZK_OK;
}

ZK_API(zk_pk_shape, (zk_ctx* c, zk_pk h, uint32_t out[8]), (c, h, out)) {
    if (!c || !out) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    auto it = c->pks.find(h);
    if (it == c->pks.end()) return ZK_EINVAL;
    const Layout& lay = it->second->lay;
    const uint32_t v[8] = {lay.k, lay.ext_k, lay.n_adv, lay.n_fix, (uint32_t)lay.perm_cols.size(), lay.n_chunks, lay.n_lookups, lay.n_h};
    memcpy(out, v, sizeof(v));
    return ZK_OK;
}

ZK_API(zk_quotient, (zk_ctx* c, zk_pk h, const zk_poly* advice_ext, size_t n_advice, const zk_poly* perm_z_ext, size_t n_chunks, const zk_poly* lookup_ext, size_t n_lookups, const uint64_t beta[4], const uint64_t gamma[4], const uint64_t y[4], int divide, zk_poly out_ext), (c, h, advice_ext, n_advice, perm_z_ext, n_chunks, lookup_ext, n_lookups, beta, gamma, y, divide, out_ext)) {
    if (!c || !advice_ext || !perm_z_ext || !lookup_ext || !beta || !gamma || !y) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    auto it = c->pks.find(h);
    if (it == c->pks.end()) return ZK_EINVAL;
    zk_pk_rec* pk = it->second;
    const Layout& lay = pk->lay;
    if (pk->srs_gen != c->srs_gen) return ZK_ESTATE;
    if (n_advice != lay.n_adv || n_chunks != lay.n_chunks || n_lookups != lay.n_lookups) return ZK_EINVAL;
    const size_t N = (size_t)4 * lay.n;
    auto ext = [&](zk_poly p) -> Fr* {
        auto q = c->polys.find(p);
        return (q == c->polys.end() || q->second.n != N) ? nullptr : q->second.ptr;
    };
    QuotientCosets qc;
    for (size_t j = 0; j < n_advice; j++) qc.adv.push_back(ext(advice_ext[j]));
    for (size_t j = 0; j < n_chunks; j++) qc.z.push_back(ext(perm_z_ext[j]));
    for (size_t l = 0; l < n_lookups; l++) {
        qc.lk_a.push_back(ext(lookup_ext[3 * l]));
        qc.lk_s.push_back(ext(lookup_ext[3 * l + 1]));
        qc.lk_z.push_back(ext(lookup_ext[3 * l + 2]));
    }
    Fr* out = ext(out_ext);
    if (!out) return ZK_EINVAL;
    for (auto* v : {&qc.adv, &qc.z, &qc.lk_a, &qc.lk_s, &qc.lk_z})
        for (const Fr* p : *v)
            if (!p || p == out) return ZK_EINVAL;
    Fr b, g, yy;
    memcpy(&b, beta, 32);
    memcpy(&g, gamma, 32);
    memcpy(&yy, y, 32);
    int rc = ctx_bind(c);
    if (rc) return rc;
    if ((rc = pk_quotient(c, pk, qc, b, g, yy, divide != 0, out))) return rc;
    HIPCHK(c, aud_sync(c, c->stream));  // the argument block is reused by the next call
    return ZK_OK;
}

ZK_API(zk_pk_set_transcript_repr, (zk_ctx* c, zk_pk h, const uint64_t transcript_repr[4]), (c, h, transcript_repr)) {
    if (!c || !transcript_repr) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    auto it = c->pks.find(h);
    if (it == c->pks.end()) return ZK_EINVAL;
    Fr v;
    memcpy(&v, transcript_repr, 32);
    // a Montgomery image is < r
    for (int i = 7; i >= 0; i--) {
        if (v.v[i] != FrParams::P[i]) {
            if (v.v[i] > FrParams::P[i]) return ZK_EINVAL;
            break;
        }
        if (i == 0) return ZK_EINVAL;
    }
    it->second->transcript_repr = v;
    // the lock-step members are by-value copies of the record (pk_make_member): every key-half field that can change after
    // they were made has to be written through to them, or proofs j > 0 of the next batch would hash the stale value
    for (zk_pk_rec* m : it->second->members) m->transcript_repr = v;
    return ZK_OK;
}

// ==================================================================== prove ==

namespace {

struct Prover {
    zk_ctx* c;
    zk_pk_rec* pk;
    const Layout& lay;
    hipStream_t st;
    ChaCha20Rng rng;
    Transcript* tr;
    uint32_t n, N;
    const Fr *tw, *tw_ext;
    Fr omega, omega_inv;
    int rc = ZK_OK;
    // staged row writes: a ring of ROWS_BLOCKS blocks in the key's pinned / device staging; `rows` is this prover's own stager,
    // or — in a lock-step batch — the first prover's, so that the blinding rows of all proofs go up in one launch per phase
    struct RowStager {
        RowEntry *host = nullptr, *dev = nullptr;
        uint32_t block = 0, count = 0;
    };
    RowStager own_rows;
    RowStager* rows = &own_rows;
    bool batch_member = false;  // one of the proofs of a lock-step batch (prover_batch.h): transforms stay on the main stream

    Prover(zk_ctx* c_, zk_pk_rec* pk_, const uint8_t seed[32], Transcript* t)
        : c(c_), pk(pk_), lay(pk_->lay), st(c_->stream), rng(seed), tr(t), n(pk_->lay.n), N(4 * pk_->lay.n) {
        own_rows.host = pk_->rows_host;
        own_rows.dev = pk_->rows_dev;
    }

    bool ok() const { return rc == ZK_OK; }
    void fail(int code) {
        if (rc == ZK_OK) rc = code;
    }

    // ---- device helpers
    // Blinding rows are staged on the host and written by rows_flush() — one upload and one launch for all
    // the columns of a phase — before the first kernel that reads those columns (commit_begin_batch and
    // transforms() flush; other readers call rows_flush() themselves).
    // ---- stream audit (audit.h, ZK_OPT_STREAM_AUDIT): every enqueue of this prover names the buffers it reads and writes
    void A(std::initializer_list<const void*> r, std::initializer_list<const void*> w, const char* site) {
        if (c->audit.on) c->audit.op(st, r, w, site);
    }
    void AV(const std::vector<const void*>& r, const std::vector<const void*>& w, const char* site) {
        if (c->audit.on) c->audit.op_v(st, r.data(), r.size(), w.data(), w.size(), site);
    }
    std::vector<const void*> aud_rows;  // the columns the staged blinding rows go to (audit only)
    void set_rows(Fr* col, uint32_t first, const std::vector<Fr>& vals) {
        if (!ok()) return;
        if (vals.size() > 8) return fail(ZK_ESTATE);
        if (rows->count == ROWS_CAP) rows_flush();
        if (c->audit.on) {
            if (rows->count == 0) c->audit.host_write(rows->host + (size_t)rows->block * ROWS_CAP, "blinding rows: the host fills a staging block");
            aud_rows.push_back(col);
        }
        RowEntry& e = rows->host[(size_t)rows->block * ROWS_CAP + rows->count++];
        memcpy(e.vals, vals.data(), vals.size() * sizeof(Fr));
        e.dst = col + first;
        e.count = (uint32_t)vals.size();
        e.pad_ = 0;
    }
    void rows_flush() {
        if (!ok() || rows->count == 0) return;
        RowEntry* h = rows->host + (size_t)rows->block * ROWS_CAP;
        RowEntry* d = rows->dev + (size_t)rows->block * ROWS_CAP;
        if (c->audit.on) {
            A({h}, {d}, "blinding rows: upload of the staging block");
            AV({d}, aud_rows, "blinding rows: scatter into the columns");
            aud_rows.clear();
        }
        if (hipMemcpyAsync(d, h, rows->count * sizeof(RowEntry), hipMemcpyHostToDevice, st) != hipSuccess) return fail(ZK_EHIP);
        launch_scatter_rows(d, rows->count, st);
        aud_record(c, c->ev_rows, st);  // what the transform stream waits for (transforms())
        rows->count = 0;
        if (++rows->block == ROWS_BLOCKS) {
            // the ring wraps: the oldest block's upload must have been consumed before it is overwritten
            if (aud_sync(c, st) != hipSuccess) return fail(ZK_EHIP);
            rows->block = 0;
        }
    }
    // commitments in flight over a set of MSM lanes, collected (written to the transcript) in the
    // order they were begun
    struct LaneFifo {
        std::vector<int> lanes;
        std::deque<int> busy;
    };
    void fifo_begin_batch(LaneFifo& f, const std::vector<const Fr*>& polys, size_t len, int basis) {
        if (!ok() || polys.empty()) return;
        if (f.busy.size() == f.lanes.size()) {
            commit_end_write(f.busy.front());
            f.busy.pop_front();
        }
        int lane = -1;
        for (int l : f.lanes)
            if (std::find(f.busy.begin(), f.busy.end(), l) == f.busy.end()) lane = l;
        commit_begin_batch(lane, polys, len, basis);
        f.busy.push_back(lane);
    }
    // gathers columns into batches of the size the MSM workspaces take; flush() launches what is pending
    struct Batcher {
        LaneFifo* f;
        int basis;
        uint32_t cap;
        std::vector<const Fr*> pend;
    };
    void batch_add(Batcher& b, const Fr* poly) {
        b.pend.push_back(poly);
        if (b.pend.size() >= b.cap) batch_flush(b);
    }
    void batch_flush(Batcher& b) {
        if (b.pend.size() >= 4 && b.f->lanes.size() >= 2 && !loaded) {
            // two passes on two lanes instead of one: the first half's reduction tail runs under the second half's head.  Only while
            // the tails have a stream of their own (a lone proof, two pipelines): under load — tails on the main stream — a second
            // pass is just a second head and tail (k = 17 EVM over four pipelines: 232.9 -> 236.9 proofs/s unsplit)
            const size_t h = (b.pend.size() + 1) / 2;
            fifo_begin_batch(*b.f, std::vector<const Fr*>(b.pend.begin(), b.pend.begin() + h), n, b.basis);
            fifo_begin_batch(*b.f, std::vector<const Fr*>(b.pend.begin() + h, b.pend.end()), n, b.basis);
        } else {
            fifo_begin_batch(*b.f, b.pend, n, b.basis);
        }
        b.pend.clear();
    }
    void fifo_drain(LaneFifo& f) {
        while (!f.busy.empty()) {
            commit_end_write(f.busy.front());
            f.busy.pop_front();
        }
    }
    std::vector<Fr> draw(uint32_t count) {
        std::vector<Fr> v(count);
        for (auto& x : v) x = rng.next_fr();
        return v;
    }
    void commit_write(const Fr* poly, size_t len, int basis) { commit_write_lane(0, poly, len, basis); }
    void commit_write_lane(int lane, const Fr* poly, size_t len, int basis) {
        commit_begin(lane, poly, len, basis);
        commit_end_write(lane);
    }
    // split commit: the MSM is enqueued on `lane` and its latency-bound tail overlaps whatever is
    // launched next; the point is written to the transcript when the lane is collected.
    void commit_begin(int lane, const Fr* poly, size_t len, int basis) { commit_begin_batch(lane, {poly}, len, basis); }
    // several columns against the same basis in ONE MSM pass (at most ctx_msm_max_batch of them)
    void commit_begin_batch(int lane, const std::vector<const Fr*>& polys, size_t len, int basis) {
        rows_flush();
        if (!ok()) return;
        int r = ctx_msm_begin_batch(c, lane, polys.data(), (uint32_t)polys.size(), basis == ZK_BASIS_LAGRANGE ? c->g_lagrange : c->g,
                                    len);
        if (r) fail(r);
    }
    // collects a lane: its commitments are written to the transcript in the order they were given
    void commit_end_write(int lane) {
        if (!ok()) return;
        G1Jac js[MSM_MAX_BATCH];
        const uint32_t cnt = c->lanes[lane].batch;
        int r = ctx_msm_end_batch(c, lane, js);
        if (r) return fail(r);
        G1Affine af[MSM_MAX_BATCH];
        jac_batch_to_affine(js, cnt, af);
        for (uint32_t q = 0; q < cnt && ok(); q++)
            if (!tr->write_point(af[q])) fail(ZK_EINVAL);  // identity: halo2 refuses to write it
    }
    // Lagrange values -> coefficients -> extended coset for a set of columns, several columns per launch
    struct Forms {
        const Fr* val;
        Fr* poly;
        Fr* coset;
    };
    // The transforms of a LONE proof run on the context's transform stream, beside the MSM passes instead of between them (ctx.h
    // xform_stream): they wait for the last flush of blinding rows — every transformed column has such rows, written after the
    // kernels that made it — and the quotient waits for them (xform_join).
    // Auto: a lone context and columns of 2^18 rows or more (measured, tools/single_ab.py OPTS=8=1 / 8=2, same box: k = 19 12.10 ->
    // 11.47 ms, EVM 13.66 -> 13.03; k = 17 6.55 -> 6.67: there the transforms are too short to pay for the cross-stream events)
    // Decided ONCE per proof (begin()): a proof whose transforms changed streams half-way would have the two streams' NTTs
    // share the context's ping-pong scratch without an order between them (round 5's first form decided per call: under four
    // pipelines the count dips to one now and then, and 1 proof in ~ 1 500 came out wrong — tools/soak.py)
    bool xside = false;
    // a quotient of three pieces (deg h < 3n) is taken over three of the extended domain's four cosets (poly.hip "three cosets"):
    // the columns' coset forms are [3][n] coset-major, made by n-point transforms.  Decided once per proof (begin())
    bool cosets3 = false;
    // three or more contexts busy on the device (the regime in which reduction tails run on the main stream): decided once per
    // proof (begin()); multi-column commitments are then not split into two passes (batch_flush)
    bool loaded = false;
    // (audit self-test, ZK_OPT_STREAM_AUDIT = 2: round 5's faulty form on purpose — the stream chosen per CALL, alternating, and no
    // join before a main-stream transform: the ledger must then refuse every proof whose transforms come in more than one call)
    bool fault_flip = false;
    bool xform_side() {
        if (c->audit_fault) return fault_flip = !fault_flip;
        return xside;
    }
    void xform_join() {
        if (!c->xform_pending) return;
        c->xform_pending = false;
        if (aud_wait(c, st, c->ev_xform) != hipSuccess) fail(ZK_EHIP);
    }
    void transforms(const std::vector<Forms>& cols) {
        rows_flush();
        if (!ok() || cols.empty()) return;
        const bool side = xform_side();
        const hipStream_t xs = side ? c->xform_stream : st;
        if (side && aud_wait(c, xs, c->ev_rows) != hipSuccess) return fail(ZK_EHIP);
        if (!side && !c->audit_fault) xform_join();  // (the two streams share the NTT's ping-pong scratch)
        transforms_on(cols, xs);
        if (side) {
            if (aud_record(c, c->ev_xform, xs) != hipSuccess) return fail(ZK_EHIP);
            c->xform_pending = true;
        }
    }
    void transforms_on(const std::vector<Forms>& cols, hipStream_t xs) {
        const uint32_t b1 = ctx_ntt_max_batch(lay.k), b2 = ctx_ntt_max_batch(lay.ext_k);
        const Fr* src[NTT_MAX_BATCH];
        Fr* dst[NTT_MAX_BATCH];
        for (size_t i0 = 0; i0 < cols.size() && ok(); i0 += b1) {
            const uint32_t cnt = (uint32_t)std::min<size_t>(b1, cols.size() - i0);
            for (uint32_t q = 0; q < cnt; q++) {
                src[q] = cols[i0 + q].val;
                dst[q] = cols[i0 + q].poly;
            }
            int r = ctx_ntt_batch(c, src, n, dst, cnt, lay.k, true, false, n, xs);
            if (r) fail(r);
        }
        if (cosets3) {
            // a quotient of three pieces: three n-point transforms per column into [3][n] coset-major values (poly.hip "three cosets")
            const uint32_t b3 = std::max(1u, b1 / 3);
            for (size_t i0 = 0; i0 < cols.size() && ok(); i0 += b3) {
                const uint32_t cnt = (uint32_t)std::min<size_t>(b3, cols.size() - i0);
                for (uint32_t q = 0; q < cnt; q++) {
                    src[q] = cols[i0 + q].poly;
                    dst[q] = cols[i0 + q].coset;
                }
                int r = ctx_ntt_cosets3(c, src, dst, cnt, lay.k, xs);
                if (r) fail(r);
            }
            return;
        }
        for (size_t i0 = 0; i0 < cols.size() && ok(); i0 += b2) {
            const uint32_t cnt = (uint32_t)std::min<size_t>(b2, cols.size() - i0);
            for (uint32_t q = 0; q < cnt; q++) {
                src[q] = cols[i0 + q].poly;
                dst[q] = cols[i0 + q].coset;
            }
            int r = ctx_ntt_batch(c, src, n, dst, cnt, lay.ext_k, false, true, N, xs);
            if (r) fail(r);
        }
    }
    // out = sum_j c_j * in_j (- sub0 on coefficient 0), any number of inputs: MAX_LC per launch
    struct Term {
        const Fr* poly;
        Fr c;
    };
    void lincomb_many(Fr* out, const std::vector<Term>& terms, bool sub0, const Fr& sub0_val, bool accumulate_first = false,
                      const std::vector<Fr>* sub_low = nullptr) {
        if (c->audit.on) {
            std::vector<const void*> rd;
            for (auto& t : terms) rd.push_back(t.poly);
            if (accumulate_first) rd.push_back(out);
            AV(rd, {out}, "linear combination");
        }
        if (terms.size() > MAX_LC && !accumulate_first && n >= 256 && pk->lc_used + terms.size() <= pk->lc_cap) {
            // hundreds of inputs: one launch over an argument list in device memory.  The list's slots are not reused within a
            // proof (the copies are asynchronous; capacity: every opened polynomial twice, pk_alloc_workspace)
            LcTerm* h = pk->h_lc_terms + pk->lc_used;
            for (size_t j = 0; j < terms.size(); j++) h[j] = LcTerm{terms[j].poly, 0, terms[j].c};
            launch_lincomb_terms(h, pk->d_lc_terms + pk->lc_used, (uint32_t)terms.size(), out, n, sub0, sub0_val,
                                 sub_low ? sub_low->data() : nullptr, sub_low ? (uint32_t)sub_low->size() : 0u, st);
            pk->lc_used += (uint32_t)terms.size();
            return;
        }
        size_t done = 0;
        bool first = !accumulate_first;
        do {
            LincombArgs a;
            memset(&a, 0, sizeof(a));
            a.out = out;
            a.n = n;
            const size_t take = std::min<size_t>(MAX_LC, terms.size() - done);
            a.count = (uint32_t)take;
            a.accumulate = first ? 0 : 1;
            for (size_t j = 0; j < take; j++) {
                a.in[j] = terms[done + j].poly;
                a.len[j] = n;
                a.c[j] = terms[done + j].c;
                a.unit[j] = terms[done + j].c == Fr::one();
            }
            done += take;
            if (done == terms.size() && sub0) {
                a.sub0 = 1;
                a.sub0_val = sub0_val;
            }
            if (done == terms.size() && sub_low) {
                a.sub_low_n = (uint32_t)sub_low->size();
                for (size_t t = 0; t < sub_low->size(); t++) a.sub_low[t] = (*sub_low)[t];
            }
            launch_lincomb(a, st);
            first = false;
        } while (done < terms.size());
    }
    Fr xrot(const Fr& x, int r) const {
        Fr w = r >= 0 ? omega : omega_inv;
        return fe_mul(x, fr_pow(w, (uint64_t)(r >= 0 ? r : -r)));
    }
    const Fr* col_val(const Col& col) const { return col.fixed ? pk->fixed_val[col.idx] : pk->adv_val[col.idx]; }
    const Fr* col_coset(const Col& col) const { return col.fixed ? pk->fixed_coset[col.idx] : pk->adv_coset[col.idx]; }

    // h(X) on the extended coset (one lane per row, quotient.hip), divided by X^n - 1, back to coefficients:
    // the first (degree - 1) * n coefficients of h_ext are the h pieces
    int quotient(const Fr& beta, const Fr& gamma, const Fr& y) {
        QuotientCosets qc;
        for (uint32_t j = 0; j < lay.n_adv; j++) qc.adv.push_back(pk->adv_coset[j]);
        for (uint32_t ci = 0; ci < lay.n_chunks; ci++) qc.z.push_back(pk->z_coset[ci]);
        for (uint32_t l = 0; l < lay.n_lookups; l++) {
            qc.lk_a.push_back(pk->lk_ap_coset[l]);
            qc.lk_s.push_back(pk->lk_sp_coset[l]);
            qc.lk_z.push_back(pk->lk_z_coset[l]);
        }
        qc.cosets3 = cosets3;
        xform_join();  // every coset form is complete
        if (!ok()) return rc;
        if (c->audit.on) {
            std::vector<const void*> rd;
            for (auto* v : {&qc.adv, &qc.z, &qc.lk_a, &qc.lk_s, &qc.lk_z})
                for (const Fr* q : *v) rd.push_back(q);
            AV(rd, {pk->h_ext}, "quotient");
        }
        int r = pk_quotient(c, pk, qc, beta, gamma, y, true, pk->h_ext);
        if (r) return r;
        if (cosets3) return ctx_intt_cosets3(c, pk->h_ext, lay.k);
        return ctx_ntt(c, pk->h_ext, N, pk->h_ext, lay.ext_k, true, true, N);
    }

    // an opening: polynomial, rotation of the point, value
    struct Q {
        const Fr* poly;
        int rot;
        Fr eval;
    };

    // every opened value of a proof in transcript order, then h(x) (not written); where the groups start
    struct EvIdx {
        size_t i_fix = 0, i_rand = 0, i_sig = 0, i_z = 0, i_lk = 0, n_written = 0;
    };
    void build_evals(std::vector<Q>& ev, EvIdx& ix) const {
        for (auto& aq : lay.advice_queries) ev.push_back(Q{pk->adv_poly[aq.first], aq.second, Fr::zero()});
        ix.i_fix = ev.size();
        for (uint32_t f = 0; f < lay.n_fix; f++) ev.push_back(Q{pk->fixed_poly[f], 0, Fr::zero()});
        ix.i_rand = ev.size();
        ev.push_back(Q{pk->random_poly, 0, Fr::zero()});
        ix.i_sig = ev.size();
        for (uint32_t p = 0; p < lay.perm_cols.size(); p++) ev.push_back(Q{pk->sigma_poly[p], 0, Fr::zero()});
        ix.i_z = ev.size();
        for (uint32_t ci = 0; ci < lay.n_chunks; ci++) {
            ev.push_back(Q{pk->z_poly[ci], 0, Fr::zero()});
            ev.push_back(Q{pk->z_poly[ci], 1, Fr::zero()});
            if (ci != lay.n_chunks - 1) ev.push_back(Q{pk->z_poly[ci], lay.last_rot, Fr::zero()});
        }
        ix.i_lk = ev.size();
        for (uint32_t l = 0; l < lay.n_lookups; l++) {
            ev.push_back(Q{pk->lk_z_poly[l], 0, Fr::zero()});
            ev.push_back(Q{pk->lk_z_poly[l], 1, Fr::zero()});
            ev.push_back(Q{pk->lk_ap_poly[l], 0, Fr::zero()});
            ev.push_back(Q{pk->lk_ap_poly[l], -1, Fr::zero()});
            ev.push_back(Q{pk->lk_sp_poly[l], 0, Fr::zero()});
        }
        ix.n_written = ev.size();
        ev.push_back(Q{pk->h_comb, 0, Fr::zero()});
    }
    // prover query order (== verifier's): advice, perm z (x, wx per chunk; then `last` in reverse), lookups
    // (zL@x, a'@x, s'@x, a'@w^-1 x, zL@wx), fixed, sigma, h, random
    std::vector<Q> queries_from_evals(const std::vector<Q>& ev, const EvIdx& ix) const {
        std::vector<Q> queries(ev.begin(), ev.begin() + ix.i_fix);
        std::vector<Q> lastq(lay.n_chunks);
        size_t pos = ix.i_z;
        for (uint32_t ci = 0; ci < lay.n_chunks; ci++) {
            queries.push_back(ev[pos++]);
            queries.push_back(ev[pos++]);
            if (ci != lay.n_chunks - 1) lastq[ci] = ev[pos++];
        }
        for (int ci = (int)lay.n_chunks - 2; ci >= 0; ci--) queries.push_back(lastq[ci]);
        pos = ix.i_lk;
        for (uint32_t l = 0; l < lay.n_lookups; l++, pos += 5) {
            queries.push_back(ev[pos]);      // zL @ x
            queries.push_back(ev[pos + 2]);  // a' @ x
            queries.push_back(ev[pos + 4]);  // s' @ x
            queries.push_back(ev[pos + 3]);  // a' @ w^-1 x
            queries.push_back(ev[pos + 1]);  // zL @ w x
        }
        for (size_t i = ix.i_fix; i < ix.i_rand; i++) queries.push_back(ev[i]);
        for (size_t i = ix.i_sig; i < ix.i_z; i++) queries.push_back(ev[i]);
        queries.push_back(ev[ix.n_written]);  // h
        queries.push_back(ev[ix.i_rand]);     // random poly
        return queries;
    }
    // h(X) = sum_i x^(n i) h_i(X) -> h_comb
    void combine_h(const Fr& x) {
        const Fr xn = fr_pow(x, n);
        LincombArgs a;
        memset(&a, 0, sizeof(a));
        a.out = pk->h_comb;
        a.n = n;
        a.count = lay.n_h;
        Fr p = Fr::one();
        for (uint32_t i = 0; i < lay.n_h; i++) {
            a.in[i] = pk->h_ext + (size_t)i * n;
            a.len[i] = n;
            a.c[i] = p;
            a.unit[i] = i == 0;
            p = fe_mul(p, xn);
        }
        A({pk->h_ext}, {pk->h_comb}, "h(X) from its pieces");
        launch_lincomb(a, st);
    }

    // ------------------------------------------------------------------ run ---
    // domain constants, and the transcript's first word
    int begin() {
        xside = !batch_member && (c->opt_xform_stream == 1 || (c->opt_xform_stream == 0 && lay.k >= 18 && ctx_activity_touch(c) <= 1));
        // the MSM passes of a lone proof on the context's MSM stream (ctx.h): same rule, same once-per-proof decision; measured
        // (tools/single_ab.py OPTS=9=1 / 9=2, four alternations on one box): 11.40-11.55 -> 11.29-11.38 ms, same bytes
        c->msm_side = !batch_member && (c->opt_msm_stream == 1 || (c->opt_msm_stream == 0 && xside && c->opt_xform_stream == 0));
        if ((xside || c->msm_side || c->audit_fault) && (rc = ctx_lone_streams(c))) return rc;
        if ((rc = ctx_get_twiddles(c, lay.k, &tw)) || (rc = ctx_get_twiddles(c, lay.ext_k, &tw_ext))) return rc;
        // auto: columns of 2^16 rows or more (measured, tools/r6_cosets3_ab.sh: k = 18 / 17 / 16 - 1 / - 3 / - 3 %, k = 17 EVM over four
        // pipelines 196 -> 203 proofs/s; the many-column rows lose — three vectors per column fill the transforms' launches three
        // times as fast: k = 13 / 12 / 11 + 5 / + 5 / + 12 %)
        // (the members of a lock-step batch get the key's coset-major copies from the batch driver: prover_batch.h)
        {
            const uint32_t above = c->opt_tail_main_above ? c->opt_tail_main_above : 2u;
            loaded = c->opt_tail_stream == 2 || (c->opt_tail_stream == 0 && (uint32_t)ctx_activity_touch(c) > above);
        }
        cosets3 = lay.n_h == 3 && 3 <= ctx_ntt_max_batch(lay.k) &&
                  (c->opt_quotient_domain == 2 || (c->opt_quotient_domain == 0 && lay.k >= 16));
        if (cosets3 && !pk->is_member && (rc = pk_ensure_cosets3(c, pk))) return rc;
        omega = fr_omega(lay.k);
        omega_inv = fe_inv_fast(omega);
        tr->common_scalar(pk->transcript_repr);
        return ZK_OK;
    }

    int run(const Fr* const* advice_dev, int scheme) {
        if (begin()) return rc;
        const uint32_t bf = BLINDING_FACTORS, usable = lay.usable;

        // Commitments are computed as early as their inputs exist (none of a', s', the random
        // polynomial needs a challenge) and collected in transcript order; RNG draws keep
        // halo2's order (the random polynomial's block range is reserved up front).
        // -- 1. advice
        // (many columns: one launch copies them all — the argument staging is reused by the later batched launches, each
        // preceded by a stream-ordered upload, so the host must not overwrite it before the upload has been consumed)
        const bool many = lay.n_adv > BATCH_ARGS_MIN;
        if (many) {
            CopyPair* h = static_cast<CopyPair*>(pk->h_batch_args);
            for (uint32_t j = 0; j < lay.n_adv; j++) h[j] = CopyPair{advice_dev[j], pk->adv_val[j]};
            if (c->audit.on) {
                std::vector<const void*> rd, wr;
                for (uint32_t j = 0; j < lay.n_adv; j++) {
                    rd.push_back(advice_dev[j]);
                    wr.push_back(pk->adv_val[j]);
                }
                AV(rd, wr, "advice columns into the workspace");
            }
            if (hipMemcpyAsync(pk->d_batch_args, h, lay.n_adv * sizeof(CopyPair), hipMemcpyHostToDevice, st) != hipSuccess) return ZK_EHIP;
            launch_copy_columns(static_cast<const CopyPair*>(pk->d_batch_args), lay.n_adv, n, st);
            if (aud_sync(c, st) != hipSuccess) return ZK_EHIP;
        }
        for (uint32_t j = 0; j < lay.n_adv; j++) {
            if (!many) {
                A({advice_dev[j]}, {pk->adv_val[j]}, "advice column into the workspace");
                hipMemcpyAsync(pk->adv_val[j], advice_dev[j], (size_t)n * sizeof(Fr), hipMemcpyDeviceToDevice, st);
            }
            set_rows(pk->adv_val[j], usable, draw(bf + 1));
        }
        draw(lay.n_adv);  // advice blinds (unused by KZG, still drawn)
        // few advice columns (k=19: one): pipeline them with the lookup commitments; many: plain order
        const bool pipe = lay.n_adv == 1 && lay.n_lookups == 1;
        // The coefficient and extended-coset forms the quotient needs are produced right behind each
        // commitment's head: they need no challenge, and they keep the main stream busy while the MSM tails
        // (and the host's transcript work) would otherwise leave it idle.
        const uint32_t max_batch = ctx_msm_max_batch(c);
        bool adv_transformed = false;
        if (pipe) {
            commit_begin(0, pk->adv_val[0], n, ZK_BASIS_LAGRANGE);
            if (xform_side()) {  // a lone proof: the advice column's forms are made under its own MSM pass
                transforms({Forms{pk->adv_val[0], pk->adv_poly[0], pk->adv_coset[0]}});
                adv_transformed = true;
            }
        } else {
            // several advice columns: whole batches of columns per MSM pass, up to MSM_LANES passes in flight,
            // each followed by its columns' transforms; collected in column order
            LaneFifo f{{0, 1, 2}, {}};
            for (uint32_t j0 = 0; j0 < lay.n_adv && ok(); j0 += max_batch) {
                const uint32_t j1 = std::min(lay.n_adv, j0 + max_batch);
                std::vector<const Fr*> cols;
                for (uint32_t j = j0; j < j1; j++) cols.push_back(pk->adv_val[j]);
                fifo_begin_batch(f, cols, n, ZK_BASIS_LAGRANGE);
                std::vector<Forms> fm;
                for (uint32_t j = j0; j < j1; j++) fm.push_back(Forms{pk->adv_val[j], pk->adv_poly[j], pk->adv_coset[j]});
                transforms(fm);
            }
            fifo_drain(f);
        }
        if (!ok()) return rc;

        // -- 2. lookups: permuted input / table (single-expression lookups: theta-compression is the identity)
        const uint32_t T = 1u << lay.lookup_bits;
        Fr theta = Fr::zero();
        bool theta_done = false;
        auto squeeze_theta = [&]() {
            if (!theta_done) {
                if (pipe) commit_end_write(0);
                theta = tr->squeeze();
                theta_done = true;
            }
        };
        hipMemsetAsync(pk->lks.err, 0, 4, st);
        // a', s' of every lookup: batches of columns per MSM pass; their transforms (and, in the pipelined
        // case, the advice column's) follow the MSM heads so that they cover the tails
        LaneFifo lf{pipe ? std::vector<int>{1, 2} : std::vector<int>{0, 1, 2}, {}};
        Batcher lb{&lf, ZK_BASIS_LAGRANGE, max_batch, {}};
        std::vector<uint32_t> due;
        auto lookup_transforms = [&](bool with_advice) {
            std::vector<Forms> fm;
            if (with_advice) fm.push_back(Forms{pk->adv_val[0], pk->adv_poly[0], pk->adv_coset[0]});
            for (uint32_t l : due) {
                fm.push_back(Forms{pk->lk_ap[l], pk->lk_ap_poly[l], pk->lk_ap_coset[l]});
                fm.push_back(Forms{pk->lk_sp[l], pk->lk_sp_poly[l], pk->lk_sp_coset[l]});
            }
            transforms(fm);
            due.clear();
        };
        if (!pipe) squeeze_theta();  // the advice commitments are all written: theta precedes the first a'
        {
            // every lookup's permuted input / table pair in one set of launches (blockIdx.y = lookup)
            LkPtrs lp;
            memset(&lp, 0, sizeof(lp));
            for (uint32_t l = 0; l < lay.n_lookups; l++) {
                if (lay.single) {
                    A({pk->adv_val[0]}, {pk->lk_in[l]}, "lookup input = q_lookup x advice");
                    launch_mul(pk->lk_in[l], pk->fixed_val[lay.fx_qlookup], pk->adv_val[0], n, st);
                    lp.inp[l] = pk->lk_in[l];
                } else {
                    lp.inp[l] = pk->adv_val[lay.n_gate + l];
                }
                lp.ap[l] = pk->lk_ap[l];
                lp.sp[l] = pk->lk_sp[l];
            }
            if (c->audit.on) {
                std::vector<const void*> rd, wr;
                for (uint32_t l = 0; l < lay.n_lookups; l++) {
                    rd.push_back(lp.inp[l]);
                    wr.push_back(lp.ap[l]);
                    wr.push_back(lp.sp[l]);
                }
                AV(rd, wr, "lookup permutation");
            }
            launch_lookup_permute(lp, lay.n_lookups, usable, T, pk->lks, st);
        }
        for (uint32_t l = 0; l < lay.n_lookups && ok(); l++) {
            set_rows(pk->lk_ap[l], usable, draw(bf + 1));
            set_rows(pk->lk_sp[l], usable, draw(bf + 1));
            draw(2);
            batch_add(lb, pk->lk_ap[l]);
            batch_add(lb, pk->lk_sp[l]);
            due.push_back(l);
            if (lb.pend.empty()) lookup_transforms(false);
        }
        batch_flush(lb);
        lookup_transforms(pipe && !adv_transformed);
        {
            // one check for all lookups (the flag accumulates): an input outside the table is halo2's
            // ConstraintSystemFailure; nothing has been written for the lookups yet
            uint32_t* err = reinterpret_cast<uint32_t*>(c->host_small);
            if (hipMemcpyAsync(err, pk->lks.err, 4, hipMemcpyDeviceToHost, st) != hipSuccess || aud_sync(c, st) != hipSuccess)
                return ZK_EHIP;
            if (*err) {
                ctx_msm_drain(c);
                return ZK_EWITNESS;
            }
        }
        squeeze_theta();
        fifo_drain(lf);
        squeeze_theta();
        (void)theta;
        if (!ok()) return rc;
        const Fr beta = tr->squeeze();
        const Fr gamma = tr->squeeze();

        // -- 5 (early). vanishing argument: the random polynomial does not depend on any challenge.
        // Its n draws come after the grand products' draws in halo2's order: reserve that block range.
        {
            const uint64_t skip = (uint64_t)lay.n_chunks * (bf + 1) + (uint64_t)lay.n_lookups * (bf + 1);
            ChaChaKey key;
            memcpy(key.w, rng.key, 32);
            A({}, {pk->random_poly}, "random polynomial");
            launch_chacha_fr(key, rng.block + skip, pk->random_poly, n, st);
            commit_begin(0, pk->random_poly, n, ZK_BASIS_MONOMIAL);
        }

        // -- 3. permutation grand products.  All z columns (permutation chunks, then lookups) are committed in
        // batches on lanes 1 and 2; their transforms follow each batch's MSM head.
        LaneFifo zf{{1, 2}, {}};
        Batcher zb{&zf, ZK_BASIS_LAGRANGE, max_batch, {}};
        std::vector<Forms> zdue;
        auto z_transforms = [&]() {
            transforms(zdue);
            zdue.clear();
        };
        {
            // numerators / denominators of every product, then all scans in one batch; a zero denominator
            // anywhere (or ZKMI355_BATCH_INVERT=1) sends every product down the batch-inversion path
            const uint32_t nprod = lay.n_chunks + lay.n_lookups;
            std::vector<GpItem> items(nprod);
            std::vector<Fr*> zs;
            const uint32_t nblk = gp_blocks(n);
            const Fr delta = fr_delta();
            Fr dcur = Fr::one();
            const bool many_chunks = lay.n_chunks > BATCH_ARGS_MIN;
            if (many_chunks && aud_sync(c, st) != hipSuccess) return ZK_EHIP;  // the argument staging may still be in use
            for (uint32_t ci = 0; ci < lay.n_chunks; ci++) {
                PermArgs a;
                memset(&a, 0, sizeof(a));
                a.n = n;
                const uint32_t lo = ci * lay.chunk_len, hi = std::min<uint32_t>((uint32_t)lay.perm_cols.size(), lo + lay.chunk_len);
                a.ncols = hi - lo;
                for (uint32_t p = lo; p < hi; p++) {
                    a.values[p - lo] = col_val(lay.perm_cols[p]);
                    a.sigma[p - lo] = pk->sigma_val[p];
                    a.delta[p - lo] = dcur;
                    dcur = fe_mul(dcur, delta);
                }
                a.tw = tw;
                a.beta = beta;
                a.gamma = gamma;
                a.num = pk->gp_num[ci];
                a.den = pk->gp_den[ci];
                if (c->audit.on) {
                    std::vector<const void*> rd;
                    for (uint32_t q = 0; q < a.ncols; q++) rd.push_back(a.values[q]);
                    AV(rd, {a.num, a.den}, "permutation numerators / denominators");  // (the batched form launches below, same stream)
                }
                if (many_chunks) static_cast<PermArgs*>(pk->h_batch_args)[ci] = a;
                else launch_perm_numden(a, st);
                zs.push_back(pk->z_val[ci]);
            }
            if (many_chunks) {
                if (hipMemcpyAsync(pk->d_batch_args, pk->h_batch_args, lay.n_chunks * sizeof(PermArgs), hipMemcpyHostToDevice, st) != hipSuccess)
                    return ZK_EHIP;
                launch_perm_numden_batch(static_cast<const PermArgs*>(pk->d_batch_args), lay.n_chunks, n, st);
            }
            const bool many_lookups = lay.n_lookups > BATCH_ARGS_MIN;
            if (many_lookups && aud_sync(c, st) != hipSuccess) return ZK_EHIP;  // the staging is rewritten below
            for (uint32_t l = 0; l < lay.n_lookups; l++) {
                const Fr* inp = lay.single ? pk->lk_in[l] : pk->adv_val[lay.n_gate + l];
                const uint32_t p = lay.n_chunks + l;
                A({pk->lk_ap[l], pk->lk_sp[l], inp}, {pk->gp_num[p], pk->gp_den[p]}, "lookup numerators / denominators");
                if (many_lookups)
                    static_cast<LkNumDenArgs*>(pk->h_batch_args)[l] =
                        LkNumDenArgs{pk->lk_ap[l], pk->lk_sp[l], inp, pk->fixed_val[lay.fx_table], pk->gp_num[p], pk->gp_den[p]};
                else
                    launch_lk_numden(pk->lk_ap[l], pk->lk_sp[l], inp, pk->fixed_val[lay.fx_table], beta, gamma, pk->gp_num[p], pk->gp_den[p], n,
                                     st);
                zs.push_back(pk->lk_z[l]);
            }
            if (many_lookups) {
                if (hipMemcpyAsync(pk->d_batch_args, pk->h_batch_args, lay.n_lookups * sizeof(LkNumDenArgs), hipMemcpyHostToDevice, st) !=
                    hipSuccess)
                    return ZK_EHIP;
                launch_lk_numden_batch(static_cast<const LkNumDenArgs*>(pk->d_batch_args), lay.n_lookups, beta, gamma, n, st);
            }
            for (uint32_t p = 0; p < nprod; p++) {
                items[p].num = pk->gp_num[p];
                items[p].den = pk->gp_den[p];
                items[p].loc_p = pk->gp_loc_p[p];
                items[p].loc_r = pk->gp_loc_r[p];
                items[p].tot_p = pk->gp_tot + (size_t)2 * nblk * p;
                items[p].tot_r = items[p].tot_p + nblk;
                items[p].z = zs[p];
                items[p].chain = (p > 0 && p < lay.n_chunks) ? 1u : 0u;  // chunk ci starts from chunk ci-1's z at row `usable`
                items[p].pad_ = 0;
            }
            Fr* q_dev = pk->gp_scal;
            Fr* qinv_dev = pk->gp_scal + nprod;
            Fr* k_dev = pk->gp_scal + 2 * (size_t)nprod;
            Fr* init_dev = pk->gp_scal + 3 * (size_t)nprod;
            if (c->audit.on) {
                std::vector<const void*> rd, wr;
                for (uint32_t p = 0; p < nprod; p++) {
                    rd.push_back(pk->gp_num[p]);
                    rd.push_back(pk->gp_den[p]);
                    wr.push_back(zs[p]);
                }
                AV(rd, wr, "grand products");  // (scan + apply, or the batch_invert fallback: same buffers, same stream)
            }
            bool fast = !c->opt_gp_batch_invert;  // zk_ctx_set_option(ZK_OPT_GP_BATCH_INVERT)
            if (fast) {
                if (hipMemcpyAsync(pk->d_gp_items, items.data(), nprod * sizeof(GpItem), hipMemcpyHostToDevice, st) != hipSuccess) return ZK_EHIP;
                launch_gp_batch_scan(pk->d_gp_items, nprod, n, q_dev, st);
                if (hipMemcpyAsync(pk->gp_host, q_dev, nprod * sizeof(Fr), hipMemcpyDeviceToHost, st) != hipSuccess ||
                    aud_sync(c, st) != hipSuccess)
                    return ZK_EHIP;
                // all inverses with one field inversion
                Fr* q = pk->gp_host;
                Fr* qi = pk->gp_host + nprod;
                Fr run = Fr::one();
                for (uint32_t p = 0; p < nprod && fast; p++) {
                    if (q[p].is_zero()) fast = false;
                    qi[p] = run;
                    run = fe_mul(run, q[p]);
                }
                if (fast) {
                    Fr inv = fe_inv_fast(run);
                    for (uint32_t p = nprod; p-- > 0;) {
                        const Fr t = fe_mul(inv, qi[p]);
                        inv = fe_mul(inv, q[p]);
                        qi[p] = t;
                    }
                    if (hipMemcpyAsync(qinv_dev, qi, nprod * sizeof(Fr), hipMemcpyHostToDevice, st) != hipSuccess) return ZK_EHIP;
                    launch_gp_batch_apply(pk->d_gp_items, nprod, n, usable, qinv_dev, k_dev, init_dev, st);
                }
            }
            if (!fast) {
                // a zero denominator: halo2's batch_invert semantics (0 -> 0), product by product
                for (uint32_t p = 0; p < nprod; p++) {
                    launch_frac(pk->gp_num[p], pk->gp_den[p], pk->t_frac, n, st);
                    const Fr* prev = items[p].chain ? zs[p - 1] + usable : nullptr;
                    launch_prefix_product(pk->t_frac, zs[p], n, prev, Fr::one(), pk->t_a, pk->t_small, st);
                }
            }
            // blinding rows and commitments, in halo2's order (chunks, then lookups)
            for (uint32_t p = 0; p < nprod && ok(); p++) {
                set_rows(zs[p], n - bf, draw(bf));
                draw(1);
                batch_add(zb, zs[p]);
                if (p < lay.n_chunks) zdue.push_back(Forms{pk->z_val[p], pk->z_poly[p], pk->z_coset[p]});
                else zdue.push_back(Forms{pk->lk_z[p - lay.n_chunks], pk->lk_z_poly[p - lay.n_chunks], pk->lk_z_coset[p - lay.n_chunks]});
                if (zb.pend.empty()) z_transforms();
            }
        }
        batch_flush(zb);
        z_transforms();
        fifo_drain(zf);
        if (!ok()) return rc;

        // -- 5. collect the random polynomial's commitment (its draws happen here in stream order)
        rng.block += n;
        draw(1);
        commit_end_write(0);
        if (!ok()) return rc;
        const Fr y = tr->squeeze();

        // -- 6. quotient (every coefficient / coset form was produced behind its commitment above)
        if (!ok()) return rc;
        if (int r = quotient(beta, gamma, y)) return r;
        draw(lay.n_h);  // h-piece blinds
        {
            // the h pieces are contiguous n-coefficient slices of the quotient: one MSM pass for all of them
            LaneFifo hf{{0, 1, 2}, {}};
            Batcher hb{&hf, ZK_BASIS_MONOMIAL, max_batch, {}};
            for (uint32_t i = 0; i < lay.n_h && ok(); i++) batch_add(hb, pk->h_ext + (size_t)i * n);
            batch_flush(hb);
            fifo_drain(hf);
        }
        if (!ok()) return rc;
        const Fr x = tr->squeeze();
        HT("x squeezed");

        // -- 7. evaluations: every opened value in ONE batched launch, then written in transcript order
        combine_h(x);
        std::vector<Q> ev;  // transcript order, then h(x) (not written)
        EvIdx ix;
        build_evals(ev, ix);
        if (ev.size() > pk->max_evals) return ZK_ESTATE;
        {
            EvalItem* ha = pk->h_evargs;
            for (size_t i = 0; i < ev.size(); i++) {
                ha[i].poly = ev[i].poly;
                ha[i].x = xrot(x, ev[i].rot);
            }
            if (c->audit.on) {
                std::vector<const void*> rd;
                for (size_t i = 0; i < ev.size(); i++) rd.push_back(ev[i].poly);
                AV(rd, {pk->ev_out}, "evaluations");
                A({pk->ev_out}, {pk->tail_host}, "evaluations to the host");
            }
            hipEventRecord(c->ev[ZK_T_EVAL][0], st);
            launch_eval_batch(ha, pk->d_evargs, (uint32_t)ev.size(), n, pk->ev_scratch, pk->ev_out, st);
            hipEventRecord(c->ev[ZK_T_EVAL][1], st);
            c->ev_valid[ZK_T_EVAL] = true;
            if (hipMemcpyAsync(pk->tail_host, pk->ev_out, ev.size() * sizeof(Fr), hipMemcpyDeviceToHost, st) != hipSuccess ||
                aud_sync(c, st) != hipSuccess)
                return ZK_EHIP;
            c->audit.host_read(pk->tail_host, "evaluations read by the host");
            for (size_t i = 0; i < ev.size(); i++) ev[i].eval = pk->tail_host[i];
        HT("evals on host");
        }
        for (size_t i = 0; i < ix.n_written; i++) tr->write_scalar(ev[i].eval);
        const std::vector<Q> queries = queries_from_evals(ev, ix);
        if (!ok()) return rc;

        // -- 8. multi-open
        pk->lc_used = 0;
        HT("queries built");
        return scheme == ZK_SCHEME_GWC ? open_gwc(queries, x, max_batch) : open_shplonk(queries, x);
    }

    // ---- multi-open, in stages: each stage ends with polynomials launched whose commitments the transcript needs next, so
    // that a lock-step prover of several proofs (prover_batch.h) can put the same commitment of all its proofs into ONE MSM
    // pass; open_gwc / open_shplonk below run the stages of one proof back to back.

    // GWC (ProverGWC, halo2_proofs poly/kzg/multiopen/gwc): one witness polynomial per rotation.
    // Stage 1 (squeezes v): every set's (sum v^i p_i - sum v^i e_i) / (X - point), all in one batched division; the witness
    // polynomials — to be committed in this order, no challenge in between — are returned in `wit`.
    int gwc_stage1(const std::vector<Q>& queries, const Fr& x, std::vector<const Fr*>& wit) {
        const Fr v = tr->squeeze();
        std::vector<std::pair<int, std::vector<Q>>> sets;
        for (auto& qq : queries) {
            bool found = false;
            for (auto& s : sets)
                if (s.first == qq.rot) {
                    s.second.push_back(qq);
                    found = true;
                    break;
                }
            if (!found) sets.push_back({qq.rot, {qq}});
        }
        // Buffers: the h pieces (free once h(X) has been combined) and two temporaries — GWC has at most six rotation sets.
        Fr* wbuf[6] = {pk->h_ext, pk->h_ext + n, pk->h_ext + 2 * (size_t)n, pk->h_ext + 3 * (size_t)n, pk->t_num, pk->t_den};
        if (sets.size() > 6) return ZK_ESTATE;
        size_t set_idx = 0;
        Fr pts[6];
        for (auto& s : sets) {
            std::vector<Term> terms;
            Fr pv = Fr::one(), eb = Fr::zero();
            for (auto& qq : s.second) {
                terms.push_back(Term{qq.poly, pv});
                eb = fe_add(eb, fe_mul(pv, qq.eval));
                pv = fe_mul(pv, v);
            }
            lincomb_many(wbuf[set_idx], terms, true, eb);
            pts[set_idx] = xrot(x, s.first);
            set_idx++;
            if (!ok()) return rc;
        }
        if (c->audit.on) {
            std::vector<const void*> b(wbuf, wbuf + set_idx);
            AV(b, b, "GWC: division by (X - point)");
        }
        launch_kate_division_batch(wbuf, wbuf, pts, (uint32_t)set_idx, n, pk->kd_scratch, st);
        for (size_t i = 0; i < set_idx; i++) wit.push_back(wbuf[i]);
        return rc;
    }
    int open_gwc(const std::vector<Q>& queries, const Fr& x, uint32_t max_batch) {
        std::vector<const Fr*> wit;
        if (int r = gwc_stage1(queries, x, wit)) return r;
        // the witness polynomials need no challenge in between: all of them go through one MSM pass
        LaneFifo wf{{0, 1, 2}, {}};
        Batcher wb{&wf, ZK_BASIS_MONOMIAL, max_batch, {}};
        for (const Fr* w : wit) batch_add(wb, w);
        batch_flush(wb);
        fifo_drain(wf);
        return rc;
    }

    // SHPLONK (ProverSHPLONK, poly/kzg/multiopen/shplonk)
    struct CR {  // a polynomial with its rotations (sorted by point value) and evaluations
        const Fr* poly;
        std::vector<int> rots;
        std::vector<Fr> evals;
    };
    struct RotPt {
        int rot;
        Fr pt, canon;
    };
    struct RS {
        std::vector<int> rots;     // sorted by point value (BTreeSet<Fr>)
        std::vector<size_t> coms;  // indices into `com`
    };
    struct Shplonk {  // what stage 2 needs of stage 1
        std::vector<CR> com;
        std::vector<RotPt> rot_pts;
        std::vector<RS> rsets;
        std::vector<int> all_rots;
        std::vector<std::vector<Fr>> low;  // per commitment: its remainder polynomial (degree < |rotation set|)
        Fr yc, v;
        Fr* hx = nullptr;
    };
    static const RotPt& sh_rot_pt(const Shplonk& S, int r) {
        for (auto& e : S.rot_pts)
            if (e.rot == r) return e;
        return S.rot_pts[0];  // (every rotation of a query is in the list: shplonk_stage1 fills it first)
    }
    // Stage 1 (squeezes y, v): h(X) = sum_i v^i (sum_j y^j (P_ij - R_ij)) / Z_i is launched; its commitment comes next.
    int shplonk_stage1(const std::vector<Q>& queries, const Fr& x, Shplonk& S) {
        // group commitments by their set of rotations
        std::vector<CR>& com = S.com;
        {
            std::unordered_map<const Fr*, size_t> seen;  // wide circuits open hundreds of polynomials: no linear searches here
            seen.reserve(queries.size());
            for (auto& qq : queries) {
                auto it = seen.find(qq.poly);
                if (it == seen.end()) {
                    it = seen.emplace(qq.poly, com.size()).first;
                    com.push_back(CR{qq.poly, {}, {}});
                }
                com[it->second].rots.push_back(qq.rot);
                com[it->second].evals.push_back(qq.eval);
            }
        }
        // the points, once per distinct rotation: x w^rot and its canonical image (BTreeSet<Fr> orders by the integer value)
        for (auto& qq : queries) {
            bool have = false;
            for (auto& e : S.rot_pts) have = have || e.rot == qq.rot;
            if (!have) {
                const Fr pt = xrot(x, qq.rot);
                S.rot_pts.push_back(RotPt{qq.rot, pt, fe_from_mont(pt)});
            }
        }
        auto pt_less = [&](int ra, int rb) {
            const Fr &a = sh_rot_pt(S, ra).canon, &b = sh_rot_pt(S, rb).canon;
            for (int i = 7; i >= 0; i--)
                if (a.v[i] != b.v[i]) return a.v[i] < b.v[i];
            return false;
        };
        std::vector<RS>& rsets = S.rsets;
        std::vector<int>& all_rots = S.all_rots;
        for (size_t ci = 0; ci < com.size(); ci++) {
            CR& cr = com[ci];
            // sort this commitment's (rot, eval) pairs by point
            std::vector<size_t> order(cr.rots.size());
            for (size_t i = 0; i < order.size(); i++) order[i] = i;
            std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return pt_less(cr.rots[a], cr.rots[b]); });
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
                if (rsets[si].rots == cr.rots) hit = si;  // (the last match, as before: sets are distinct, so the only one)
            if (hit == rsets.size()) rsets.push_back(RS{cr.rots, {}});
            rsets[hit].coms.push_back(ci);
        }
        std::sort(all_rots.begin(), all_rots.end(), pt_less);
        HT("grouped");
        S.yc = tr->squeeze();
        S.v = tr->squeeze();
        const Fr yc = S.yc, v = S.v;
        S.low.assign(com.size(), {});
        // h(X) = sum_i v^i * ( sum_j y^j (P_ij - R_ij) ) / Z_i.  Every rotation set has its own buffer (the h pieces
        // are free by now); step s divides, in ONE batched launch, every set that still has a point left by it.
        Fr* hx = pk->t_frac;  // h(X)
        S.hx = hx;
        Fr* sbuf[6] = {pk->h_ext, pk->h_ext + n, pk->h_ext + 2 * (size_t)n, pk->h_ext + 3 * (size_t)n, pk->t_num, pk->t_den};
        if (rsets.size() > 6) return ZK_ESTATE;
        std::vector<std::vector<Fr>> set_pts;
        size_t max_pts = 0;
        for (size_t si = 0; si < rsets.size(); si++) {
            auto& rs = rsets[si];
            std::vector<Fr> pts;
            for (int r : rs.rots) pts.push_back(sh_rot_pt(S, r).pt);
            const std::vector<std::vector<Fr>> basis = lagrange_basis(pts);
            std::vector<Term> terms;
            std::vector<Fr> rsum(pts.size(), Fr::zero());
            Fr py = Fr::one();
            for (size_t ci : rs.coms) {
                const CR& cr = com[ci];
                std::vector<Fr>& lo = S.low[ci];
                lo.assign(pts.size(), Fr::zero());
                for (size_t j = 0; j < pts.size(); j++)
                    for (size_t t = 0; t < pts.size(); t++) lo[t] = fe_add(lo[t], fe_mul(basis[j][t], cr.evals[j]));
                terms.push_back(Term{cr.poly, py});
                for (size_t t = 0; t < pts.size(); t++) rsum[t] = fe_add(rsum[t], fe_mul(py, lo[t]));
                py = fe_mul(py, yc);
            }
            // sum_j y^j P_j(X) minus sum_j y^j R_j(X) (degree < |set|: a few low coefficients, known on the host)
            if (pts.size() > 8) return ZK_ESTATE;
            lincomb_many(sbuf[si], terms, false, Fr::zero(), false, &rsum);
        HT("set lincomb launched");
            max_pts = std::max(max_pts, pts.size());
            set_pts.push_back(pts);
        }
        for (size_t step = 0; step < max_pts; step++) {
            Fr* bufs[6];
            Fr zs[6];
            uint32_t cnt = 0;
            for (size_t si = 0; si < rsets.size(); si++)
                if (step < set_pts[si].size()) {
                    bufs[cnt] = sbuf[si];
                    zs[cnt] = set_pts[si][step];
                    cnt++;
                }
            if (c->audit.on) {
                std::vector<const void*> b(bufs, bufs + cnt);
                AV(b, b, "SHPLONK: division step");
            }
            launch_kate_division_batch(bufs, bufs, zs, cnt, n, pk->kd_scratch, st);
        }
        {
            std::vector<Term> terms;
            Fr pv = Fr::one();
            for (size_t si = 0; si < rsets.size(); si++) {
                terms.push_back(Term{sbuf[si], pv});
                pv = fe_mul(pv, v);
            }
            lincomb_many(hx, terms, false, Fr::zero());
        HT("hx launched");
        }
        return rc;
    }
    // Stage 2 (h's commitment is in the transcript; squeezes u): the final quotient (L(X) / (X - u)) / z_0 is launched in
    // *out; its commitment ends the proof.
    int shplonk_stage2(Shplonk& S, const Fr** out) {
        const Fr u = tr->squeeze();
        HT("u squeezed");
        // L(X) = sum_i v^i z_i sum_j y^j (P_ij(X) - R_ij(u)) - Z_T(u) h(X)
        std::vector<Term> terms;
        Fr sub = Fr::zero();
        Fr pv = Fr::one();
        std::vector<Fr> z_diffs;
        for (auto& rs : S.rsets) {
            std::vector<Fr> diffs;
            for (int r : S.all_rots)
                if (std::find(rs.rots.begin(), rs.rots.end(), r) == rs.rots.end()) diffs.push_back(sh_rot_pt(S, r).pt);
            const Fr zi = vanishing_eval(diffs, u);
            z_diffs.push_back(zi);
            Fr py = Fr::one();
            for (size_t ci : rs.coms) {
                const Fr coef = fe_mul(fe_mul(pv, zi), py);
                terms.push_back(Term{S.com[ci].poly, coef});
                sub = fe_add(sub, fe_mul(coef, eval_small(S.low[ci], u)));
                py = fe_mul(py, S.yc);
            }
            pv = fe_mul(pv, S.v);
        }
        std::vector<Fr> all_pts;
        for (int r : S.all_rots) all_pts.push_back(sh_rot_pt(S, r).pt);
        const Fr zt = vanishing_eval(all_pts, u);
        terms.push_back(Term{S.hx, fe_neg(zt)});
        lincomb_many(pk->t_a, terms, true, sub);
        HT("L launched");
        A({pk->t_a}, {pk->t_b}, "SHPLONK: final division");
        launch_kate_division(pk->t_a, pk->t_b, n, u, pk->kd_scratch, st);
        A({pk->t_b}, {pk->t_b}, "SHPLONK: scale");
        launch_scale(pk->t_b, fe_inv_fast(z_diffs[0]), n, st);
        *out = pk->t_b;
        return rc;
    }
    int open_shplonk(const std::vector<Q>& queries, const Fr& x) {
        Shplonk S;
        if (int r = shplonk_stage1(queries, x, S)) return r;
        commit_write(S.hx, n, ZK_BASIS_MONOMIAL);
        if (!ok()) return rc;
        const Fr* last = nullptr;
        if (int r = shplonk_stage2(S, &last)) return r;
        commit_write(last, n, ZK_BASIS_MONOMIAL);
        return rc;
    }
};

#include "prover_batch.h"

}  // namespace

// Leaves the context as every entry point expects to find it — no commitment in flight, the side streams drained and their
// flags cleared, the main stream idle — on EVERY way out of a whole-proof call, including an exception thrown inside the
// prover (std::bad_alloc from its vectors: ZK_API turns it into ZK_EINTERNAL after this destructor has run)
namespace {
struct ProveQuiesce {
    zk_ctx* c;
    explicit ProveQuiesce(zk_ctx* ctx) : c(ctx) { ctx_activity_hold(c, true); }
    ProveQuiesce(const ProveQuiesce&) = delete;
    ProveQuiesce& operator=(const ProveQuiesce&) = delete;
    void settle() {
        ctx_msm_drain(c);  // an early error may leave commitments in flight
        if (c->msm_side) {
            aud_sync(c, c->msm_stream);
            c->msm_side = false;
        }
        if (c->xform_pending) {  // (an early error before the quotient: transforms still in flight)
            aud_sync(c, c->xform_stream);
            c->xform_pending = false;
        }
        aud_sync(c, c->stream);
        ctx_activity_hold(c, false);
    }
    ~ProveQuiesce() { settle(); }
};
}  // namespace

ZK_API(zk_proof_size, (zk_ctx* c, zk_pk pkh, int transcript, int scheme, size_t* out), (c, pkh, transcript, scheme, out)) {
    if (!c || !out) return ZK_EINVAL;
    std::lock_guard<std::mutex> g(c->mu);
    auto it = c->pks.find(pkh);
    if (it == c->pks.end()) return ZK_EINVAL;
    if (transcript != ZK_TRANSCRIPT_BLAKE2B && transcript != ZK_TRANSCRIPT_EVM) return ZK_EINVAL;
    if (scheme == ZK_SCHEME_DEFAULT) scheme = transcript == ZK_TRANSCRIPT_EVM ? ZK_SCHEME_GWC : ZK_SCHEME_SHPLONK;
    if (scheme != ZK_SCHEME_GWC && scheme != ZK_SCHEME_SHPLONK) return ZK_EINVAL;
    const Layout& lay = it->second->lay;
    // commitments: advice, (a', s', zL) per lookup, z per chunk, random, h pieces; then the opening proof:
    // SHPLONK two points, GWC one per distinct rotation {0,1,2,3,-1} (+ `last` once there is a chunk link)
    size_t points = lay.n_adv + 3 * lay.n_lookups + lay.n_chunks + 1 + lay.n_h;
    points += scheme == ZK_SCHEME_SHPLONK ? 2 : 5 + (lay.n_chunks > 1 ? 1 : 0);
    const size_t evals = lay.advice_queries.size() + lay.n_fix + 1 + lay.perm_cols.size() + 3 * lay.n_chunks - 1 +
                         5 * lay.n_lookups;
    *out = points * (transcript == ZK_TRANSCRIPT_EVM ? 64 : 32) + evals * 32;
    return ZK_OK;
}

ZK_API(zk_prove, (zk_ctx* c, zk_pk h, const zk_poly* advice, size_t n_advice, const uint8_t rng_seed[32], int transcript, int scheme, uint8_t* proof_out, size_t proof_cap, size_t* proof_len), (c, h, advice, n_advice, rng_seed, transcript, scheme, proof_out, proof_cap, proof_len)) {
    if (!c || !advice || !rng_seed || !proof_len) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    auto it = c->pks.find(h);
    if (it == c->pks.end()) return ZK_EINVAL;
    zk_pk_rec* pk = it->second;
    const Layout& lay = pk->lay;
    if (pk->srs_gen != c->srs_gen) return ZK_ESTATE;  // the SRS was replaced after this key was made: its vk is stale
    if (n_advice != lay.n_adv || c->srs_k != (int)lay.k) return ZK_EINVAL;
    if (transcript != ZK_TRANSCRIPT_BLAKE2B && transcript != ZK_TRANSCRIPT_EVM) return ZK_EINVAL;
    if (scheme == ZK_SCHEME_DEFAULT) scheme = transcript == ZK_TRANSCRIPT_EVM ? ZK_SCHEME_GWC : ZK_SCHEME_SHPLONK;
    if (scheme != ZK_SCHEME_GWC && scheme != ZK_SCHEME_SHPLONK) return ZK_EINVAL;
    int rc = ctx_bind(c);
    if (rc) return rc;
    std::vector<const Fr*> adv(n_advice);
    for (size_t j = 0; j < n_advice; j++) {
        auto pit = c->polys.find(advice[j]);
        if (pit == c->polys.end() || pit->second.n != lay.n) return ZK_EINVAL;
        adv[j] = pit->second.ptr;
    }
    EvmTranscript evm;
    Blake2bTranscript b2;
    Transcript* tr = transcript == ZK_TRANSCRIPT_EVM ? (Transcript*)&evm : (Transcript*)&b2;
    const uint64_t aud0 = c->audit.violations;
    c->audit.base_of.clear();  // (allocations may have changed hands since the last proof)
    Prover p(c, pk, rng_seed, tr);
    {
        ProveQuiesce quiesce(c);  // (declared after the prover: it settles the streams while the prover's host buffers are alive)
        rc = p.run(adv.data(), scheme);
    }
    if ((rc = aud_verdict(c, aud0, rc))) return rc;
    if (hipGetLastError() != hipSuccess) return ZK_EHIP;
    *proof_len = tr->out.size();
    if (!proof_out || proof_cap < tr->out.size()) return proof_out ? ZK_EINVAL : ZK_OK;
    memcpy(proof_out, tr->out.data(), tr->out.size());
    return ZK_OK;
}

// ===================================================== phase-level entry points ==
// For a host that keeps halo2's own prover flow — its transcript, its RNG, its blinding — and off-loads phase by phase
// (examples/prove_host_phases.cpp): the provers of plonk/lookup, plonk/permutation and plonk/vanishing that
// sit between the commitments, over resident columns.  They run the very kernels zk_prove runs; each call is complete on return.
namespace {
struct PhaseCtx {
    zk_ctx* c;
    zk_pk_rec* pk;
    std::vector<Fr*> adv;
};
// resolves the key and the advice handles (n rows each, Lagrange values, Montgomery)
int phase_open(zk_ctx* c, zk_pk h, const zk_poly* advice, size_t n_advice, PhaseCtx& out) {
    auto it = c->pks.find(h);
    if (it == c->pks.end()) return ZK_EINVAL;
    zk_pk_rec* pk = it->second;
    if (pk->srs_gen != c->srs_gen) return ZK_ESTATE;
    if (n_advice != pk->lay.n_adv) return ZK_EINVAL;
    out.c = c;
    out.pk = pk;
    for (size_t j = 0; j < n_advice; j++) {
        auto pit = c->polys.find(advice[j]);
        if (pit == c->polys.end() || pit->second.n != pk->lay.n) return ZK_EINVAL;
        out.adv.push_back(pit->second.ptr);
    }
    return ctx_bind(c);
}
Fr* phase_vec(zk_ctx* c, zk_poly h, size_t n) {
    auto it = c->polys.find(h);
    return (it == c->polys.end() || it->second.n != n) ? nullptr : it->second.ptr;
}
// the compressed input expression of lookup l: the lookup advice column, or q_lookup * a for the one-column shape (into pk->lk_in)
const Fr* phase_lookup_input(PhaseCtx& P, uint32_t l) {
    const Layout& lay = P.pk->lay;
    if (!lay.single) return P.adv[lay.n_gate + l];
    launch_mul(P.pk->lk_in[l], P.pk->fixed_val[lay.fx_qlookup], P.adv[0], lay.n, P.c->stream);
    return P.pk->lk_in[l];
}
// grand products z[p] of `items` (num / den already launched), halo2's semantics: one scan, one host round trip; a zero
// denominator takes the batch_invert form.  chain[p]: product p starts from product p - 1's value at row `usable`.
int phase_grand_products(zk_ctx* c, zk_pk_rec* pk, const std::vector<Fr*>& num, const std::vector<Fr*>& den, const std::vector<Fr*>& z,
                         const std::vector<uint32_t>& chain) {
    const Layout& lay = pk->lay;
    const uint32_t n = lay.n, nprod = (uint32_t)z.size(), nblk = gp_blocks(n), usable = lay.usable;
    hipStream_t st = c->stream;
    std::vector<GpItem> items(nprod);
    for (uint32_t p = 0; p < nprod; p++) {
        items[p].num = num[p];
        items[p].den = den[p];
        items[p].loc_p = pk->gp_loc_p[p];
        items[p].loc_r = pk->gp_loc_r[p];
        items[p].tot_p = pk->gp_tot + (size_t)2 * nblk * p;
        items[p].tot_r = items[p].tot_p + nblk;
        items[p].z = z[p];
        items[p].chain = chain[p];
        items[p].pad_ = 0;
    }
    Fr *q_dev = pk->gp_scal, *qinv_dev = pk->gp_scal + nprod, *k_dev = pk->gp_scal + 2 * (size_t)nprod, *init_dev = pk->gp_scal + 3 * (size_t)nprod;
    bool fast = !c->opt_gp_batch_invert;
    if (fast) {
        HIPCHK(c, hipMemcpyAsync(pk->d_gp_items, items.data(), nprod * sizeof(GpItem), hipMemcpyHostToDevice, st));
        launch_gp_batch_scan(pk->d_gp_items, nprod, n, q_dev, st);
        HIPCHK(c, hipMemcpyAsync(pk->gp_host, q_dev, nprod * sizeof(Fr), hipMemcpyDeviceToHost, st));
        HIPCHK(c, aud_sync(c, st));
        Fr *q = pk->gp_host, *qi = pk->gp_host + nprod;
        Fr run = Fr::one();
        for (uint32_t p = 0; p < nprod && fast; p++) {
            if (q[p].is_zero()) fast = false;
            qi[p] = run;
            run = fe_mul(run, q[p]);
        }
        if (fast) {
            Fr inv = fe_inv_fast(run);
            for (uint32_t p = nprod; p-- > 0;) {
                const Fr t = fe_mul(inv, qi[p]);
                inv = fe_mul(inv, q[p]);
                qi[p] = t;
            }
            HIPCHK(c, hipMemcpyAsync(qinv_dev, qi, nprod * sizeof(Fr), hipMemcpyHostToDevice, st));
            launch_gp_batch_apply(pk->d_gp_items, nprod, n, usable, qinv_dev, k_dev, init_dev, st);
        }
    }
    if (!fast) {
        for (uint32_t p = 0; p < nprod; p++) {
            launch_frac(num[p], den[p], pk->t_frac, n, st);
            const Fr* prev = chain[p] ? z[p - 1] + usable : nullptr;
            launch_prefix_product(pk->t_frac, z[p], n, prev, Fr::one(), pk->t_a, pk->t_small, st);
        }
    }
    HIPCHK(c, aud_sync(c, st));
    return hipGetLastError() == hipSuccess ? ZK_OK : ZK_EHIP;
}
}  // namespace

namespace {
// every output vector of a phase call is written by its own blocks of a batched launch: the same vector twice among the
// outputs, or an output that another item of the call reads, is a data race that yields garbage — refused before anything is
// launched
bool phase_outputs_ok(const std::vector<const Fr*>& outs, const std::vector<const Fr*>& ins) {
    for (size_t i = 0; i < outs.size(); i++) {
        for (size_t j = i + 1; j < outs.size(); j++)
            if (outs[i] == outs[j]) return false;
        for (const Fr* v : ins)
            if (outs[i] == v) return false;
    }
    return true;
}
}  // namespace

ZK_API(zk_lookup_permute, (zk_ctx* c, zk_pk h, const zk_poly* advice, size_t n_advice, zk_poly* permuted_input, zk_poly* permuted_table, size_t n_lookups), (c, h, advice, n_advice, permuted_input, permuted_table, n_lookups)) {
    if (!c || !advice || !permuted_input || !permuted_table) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    PhaseCtx P;
    int rc = phase_open(c, h, advice, n_advice, P);
    if (rc) return rc;
    const Layout& lay = P.pk->lay;
    if (n_lookups != lay.n_lookups) return ZK_EINVAL;
    LkPtrs lp;
    memset(&lp, 0, sizeof(lp));
    std::vector<const Fr*> outs, ins(P.adv.begin(), P.adv.end());
    for (uint32_t l = 0; l < lay.n_lookups; l++) {
        Fr *a = phase_vec(c, permuted_input[l], lay.n), *s = phase_vec(c, permuted_table[l], lay.n);
        if (!a || !s) return ZK_EINVAL;
        lp.ap[l] = a;
        lp.sp[l] = s;
        outs.push_back(a);
        outs.push_back(s);
    }
    if (!phase_outputs_ok(outs, ins)) return ZK_EINVAL;  // (a'[l] = s'[m], a repeated handle, an advice column as an output)
    hipStream_t st = c->stream;
    HIPCHK(c, hipMemsetAsync(P.pk->lks.err, 0, 4, st));
    for (uint32_t l = 0; l < lay.n_lookups; l++) lp.inp[l] = phase_lookup_input(P, l);
    launch_lookup_permute(lp, lay.n_lookups, lay.usable, 1u << lay.lookup_bits, P.pk->lks, st);
    uint32_t* err = reinterpret_cast<uint32_t*>(c->host_small);
    HIPCHK(c, hipMemcpyAsync(err, P.pk->lks.err, 4, hipMemcpyDeviceToHost, st));
    HIPCHK(c, aud_sync(c, st));
    return *err ? ZK_EWITNESS : ZK_OK;
}

ZK_API(zk_lookup_product, (zk_ctx* c, zk_pk h, const zk_poly* advice, size_t n_advice, const zk_poly* permuted_input, const zk_poly* permuted_table, size_t n_lookups, const uint64_t beta[4], const uint64_t gamma[4], zk_poly* z_out), (c, h, advice, n_advice, permuted_input, permuted_table, n_lookups, beta, gamma, z_out)) {
    if (!c || !advice || !permuted_input || !permuted_table || !beta || !gamma || !z_out) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    PhaseCtx P;
    int rc = phase_open(c, h, advice, n_advice, P);
    if (rc) return rc;
    zk_pk_rec* pk = P.pk;
    const Layout& lay = pk->lay;
    if (n_lookups != lay.n_lookups) return ZK_EINVAL;
    Fr b, g;
    memcpy(&b, beta, 32);
    memcpy(&g, gamma, 32);
    std::vector<Fr*> num, den, z;
    std::vector<uint32_t> chain;
    std::vector<const Fr*> outs, ins(P.adv.begin(), P.adv.end()), av, sv;
    for (uint32_t l = 0; l < lay.n_lookups; l++) {
        const Fr *a = phase_vec(c, permuted_input[l], lay.n), *s = phase_vec(c, permuted_table[l], lay.n);
        Fr* zl = phase_vec(c, z_out[l], lay.n);
        if (!a || !s || !zl) return ZK_EINVAL;
        av.push_back(a);
        sv.push_back(s);
        ins.push_back(a);
        ins.push_back(s);
        outs.push_back(zl);
        z.push_back(zl);
    }
    if (!phase_outputs_ok(outs, ins)) return ZK_EINVAL;  // (before the first launch: a repeated z, or a z that some lookup reads)
    for (uint32_t l = 0; l < lay.n_lookups; l++) {
        const Fr* inp = phase_lookup_input(P, l);
        launch_lk_numden(av[l], sv[l], inp, pk->fixed_val[lay.fx_table], b, g, pk->gp_num[l], pk->gp_den[l], lay.n, c->stream);
        num.push_back(pk->gp_num[l]);
        den.push_back(pk->gp_den[l]);
        chain.push_back(0);
    }
    return phase_grand_products(c, pk, num, den, z, chain);
}

ZK_API(zk_permutation_product, (zk_ctx* c, zk_pk h, const zk_poly* advice, size_t n_advice, const uint64_t beta[4], const uint64_t gamma[4], zk_poly* z_out, size_t n_chunks), (c, h, advice, n_advice, beta, gamma, z_out, n_chunks)) {
    if (!c || !advice || !beta || !gamma || !z_out) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    PhaseCtx P;
    int rc = phase_open(c, h, advice, n_advice, P);
    if (rc) return rc;
    zk_pk_rec* pk = P.pk;
    const Layout& lay = pk->lay;
    if (n_chunks != lay.n_chunks) return ZK_EINVAL;
    Fr b, g;
    memcpy(&b, beta, 32);
    memcpy(&g, gamma, 32);
    const Fr* tw = nullptr;
    if ((rc = ctx_get_twiddles(c, lay.k, &tw))) return rc;
    std::vector<Fr*> num, den, z;
    std::vector<uint32_t> chain;
    const Fr delta = fr_delta();
    Fr dcur = Fr::one();
    {
        std::vector<const Fr*> outs, ins(P.adv.begin(), P.adv.end());
        for (uint32_t ci = 0; ci < lay.n_chunks; ci++) {
            const Fr* zc = phase_vec(c, z_out[ci], lay.n);
            if (!zc) return ZK_EINVAL;
            outs.push_back(zc);
        }
        if (!phase_outputs_ok(outs, ins)) return ZK_EINVAL;  // (before the first launch: a repeated z, an advice column as z)
    }
    for (uint32_t ci = 0; ci < lay.n_chunks; ci++) {
        Fr* zc = phase_vec(c, z_out[ci], lay.n);
        PermArgs a;
        memset(&a, 0, sizeof(a));
        a.n = lay.n;
        const uint32_t lo = ci * lay.chunk_len, hi = std::min<uint32_t>((uint32_t)lay.perm_cols.size(), lo + lay.chunk_len);
        a.ncols = hi - lo;
        for (uint32_t p = lo; p < hi; p++) {
            const Col& col = lay.perm_cols[p];
            a.values[p - lo] = col.fixed ? pk->fixed_val[col.idx] : P.adv[col.idx];
            a.sigma[p - lo] = pk->sigma_val[p];
            a.delta[p - lo] = dcur;
            dcur = fe_mul(dcur, delta);
        }
        a.tw = tw;
        a.beta = b;
        a.gamma = g;
        a.num = pk->gp_num[ci];
        a.den = pk->gp_den[ci];
        launch_perm_numden(a, c->stream);
        num.push_back(pk->gp_num[ci]);
        den.push_back(pk->gp_den[ci]);
        z.push_back(zc);
        chain.push_back(ci > 0 ? 1u : 0u);
    }
    return phase_grand_products(c, pk, num, den, z, chain);
}

// a copy of one of the key's own polynomials (coefficient form) in a caller's vector: what a phase-driving host evaluates and
// opens beside its own columns (the fixed and permutation polynomials of the ProvingKey)
ZK_API(zk_pk_export_poly, (zk_ctx* c, zk_pk h, int which, size_t index, zk_poly dst), (c, h, which, index, dst)) {
    if (!c) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    auto it = c->pks.find(h);
    if (it == c->pks.end()) return ZK_EINVAL;
    zk_pk_rec* pk = it->second;
    if (pk->srs_gen != c->srs_gen) return ZK_ESTATE;
    const std::vector<Fr*>* v = which == ZK_PK_FIXED_POLY ? &pk->fixed_poly : which == ZK_PK_SIGMA_POLY ? &pk->sigma_poly : nullptr;
    if (!v || index >= v->size()) return ZK_EINVAL;
    Fr* d = phase_vec(c, dst, pk->lay.n);
    if (!d) return ZK_EINVAL;
    int rc = ctx_bind(c);
    if (rc) return rc;
    HIPCHK(c, hipMemcpyAsync(d, (*v)[index], (size_t)pk->lay.n * sizeof(Fr), hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(c, aud_sync(c, c->stream));
    return ZK_OK;
}

// the vanishing argument's random polynomial: coefficient i = Fr::random of ChaCha20 block first_block + i under `key` — the
// stream `ChaCha20Rng::from_seed(key)` yields when every draw is an Fr::random (one 64-byte block each), i.e. what the host's
// RNG would give for draws first_block .. first_block + n - 1; the host then advances its own RNG by n draws
ZK_API(zk_random_poly, (zk_ctx* c, const uint8_t chacha_key[32], uint64_t first_block, zk_poly out), (c, chacha_key, first_block, out)) {
    if (!c || !chacha_key) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    auto it = c->polys.find(out);
    if (it == c->polys.end() || it->second.n == 0 || it->second.n > ((size_t)1 << 28)) return ZK_EINVAL;
    int rc = ctx_bind(c);
    if (rc) return rc;
    ChaChaKey key;
    memcpy(key.w, chacha_key, 32);
    launch_chacha_fr(key, first_block, it->second.ptr, (uint32_t)it->second.n, c->stream);
    HIPCHK(c, aud_sync(c, c->stream));
    return ZK_OK;
}

// out = sum_j coeffs[j] * in[j] - (sub_low[0] + sub_low[1] X + ..): the multi-open provers' linear combinations (GWC subtracts the
// combined evaluation, SHPLONK the combined remainder polynomial of a rotation set) and h(X) = sum x^(n i) h_i
ZK_API(zk_poly_lincomb, (zk_ctx* c, zk_poly out, const zk_poly* in, const uint64_t* coeffs, size_t count, const uint64_t* sub_low, size_t n_low), (c, out, in, coeffs, count, sub_low, n_low)) {
    if (!c || !in || !coeffs || count == 0 || n_low > 8 || (n_low && !sub_low)) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    auto oit = c->polys.find(out);
    if (oit == c->polys.end() || oit->second.n > 0xffffffffu) return ZK_EINVAL;
    const size_t n = oit->second.n;
    std::vector<const Fr*> src(count);
    for (size_t j = 0; j < count; j++) {
        auto it = c->polys.find(in[j]);
        if (it == c->polys.end() || it->second.n != n || it->second.ptr == oit->second.ptr) return ZK_EINVAL;
        src[j] = it->second.ptr;
    }
    int rc = ctx_bind(c);
    if (rc) return rc;
    size_t done = 0;
    bool first = true;
    do {
        LincombArgs a;
        memset(&a, 0, sizeof(a));
        a.out = oit->second.ptr;
        a.n = (uint32_t)n;
        const size_t take = std::min<size_t>(MAX_LC, count - done);
        a.count = (uint32_t)take;
        a.accumulate = first ? 0 : 1;
        for (size_t j = 0; j < take; j++) {
            a.in[j] = src[done + j];
            a.len[j] = (uint32_t)n;
            memcpy(&a.c[j], coeffs + 4 * (done + j), 32);
            a.unit[j] = a.c[j] == Fr::one();
        }
        done += take;
        if (done == count && n_low) {  // the low-degree polynomial subtracted from the first n_low coefficients
            a.sub_low_n = (uint32_t)n_low;
            memcpy(a.sub_low, sub_low, n_low * 32);
        }
        launch_lincomb(a, c->stream);
        first = false;
    } while (done < count);
    HIPCHK(c, aud_sync(c, c->stream));
    return ZK_OK;
}

// create_proof for `batch` independent proofs of one key in lock-step (prover_batch.h)
ZK_API(zk_prove_batch, (zk_ctx* c, zk_pk h, size_t batch, const zk_poly* advice, size_t n_advice, const uint8_t* rng_seeds, int transcript, int scheme, uint8_t* proofs_out, size_t proof_stride, size_t* proof_len), (c, h, batch, advice, n_advice, rng_seeds, transcript, scheme, proofs_out, proof_stride, proof_len)) {
    if (!c || !advice || !rng_seeds || !proof_len || batch == 0 || batch > ZK_PROVE_BATCH_MAX) return ZK_EINVAL;
    if (batch == 1) return zk_prove(c, h, advice, n_advice, rng_seeds, transcript, scheme, proofs_out, proof_stride, proof_len);
    std::lock_guard<std::mutex> lk(c->mu);
    auto it = c->pks.find(h);
    if (it == c->pks.end()) return ZK_EINVAL;
    zk_pk_rec* pk = it->second;
    const Layout& lay = pk->lay;
    if (pk->srs_gen != c->srs_gen) return ZK_ESTATE;  // the SRS was replaced after this key was made: its vk is stale
    if (n_advice != lay.n_adv || c->srs_k != (int)lay.k) return ZK_EINVAL;
    if (transcript != ZK_TRANSCRIPT_BLAKE2B && transcript != ZK_TRANSCRIPT_EVM) return ZK_EINVAL;
    if (scheme == ZK_SCHEME_DEFAULT) scheme = transcript == ZK_TRANSCRIPT_EVM ? ZK_SCHEME_GWC : ZK_SCHEME_SHPLONK;
    if (scheme != ZK_SCHEME_GWC && scheme != ZK_SCHEME_SHPLONK) return ZK_EINVAL;
    const uint32_t B = (uint32_t)batch;
    // all grand products of the batch are scanned by one 256-lane workgroup (gp_chain_kernel)
    if ((uint64_t)B * (lay.n_chunks + lay.n_lookups) > 256) return ZK_EINVAL;
    int rc = ctx_bind(c);
    if (rc) return rc;
    std::vector<const Fr*> adv(batch * n_advice);
    for (size_t j = 0; j < batch * n_advice; j++) {
        auto pit = c->polys.find(advice[j]);
        if (pit == c->polys.end() || pit->second.n != lay.n) return ZK_EINVAL;
        adv[j] = pit->second.ptr;
    }
    if ((rc = pk_ensure_batch(c, pk, B))) return rc;
    // columns per MSM pass: the same commitment of all proofs at once, two columns per proof where a phase has them (a', s';
    // z, zL) — ZK_OPT_BATCH_PASS_COLUMNS overrides; the lanes' workspaces grow to that on their next pass
    uint32_t cap = c->opt_batch_pass_cols ? c->opt_batch_pass_cols : std::max(std::min<uint32_t>(2 * B, 8u), ctx_msm_max_batch(c));
    cap = std::min<uint32_t>(cap, MSM_MAX_BATCH);
    if (!c->table_c) cap = 1;  // no window tables (k < 10): one column per pass
    c->msm_min_cols = std::max(c->msm_min_cols, cap);
    std::vector<std::unique_ptr<Transcript>> trs;
    std::vector<std::unique_ptr<Prover>> provers;
    std::vector<Prover*> P;
    for (uint32_t q = 0; q < B; q++) {
        trs.emplace_back(transcript == ZK_TRANSCRIPT_EVM ? (Transcript*)new EvmTranscript() : (Transcript*)new Blake2bTranscript());
        provers.emplace_back(new Prover(c, q == 0 ? pk : pk->members[q - 1], rng_seeds + 32 * (size_t)q, trs.back().get()));
        P.push_back(provers.back().get());
    }
    const uint64_t aud0 = c->audit.violations;
    c->audit.base_of.clear();
    {
        BatchRun run(c, pk, P, cap);
        ProveQuiesce quiesce(c);
        rc = run.run(adv.data(), scheme);
    }
    if ((rc = aud_verdict(c, aud0, rc))) return rc;
    if (hipGetLastError() != hipSuccess) return ZK_EHIP;
    const size_t len = trs[0]->out.size();
    for (uint32_t q = 0; q < B; q++)
        if (trs[q]->out.size() != len) return ZK_EINTERNAL;  // (one shape, one length)
    *proof_len = len;
    if (!proofs_out) return ZK_OK;
    if (proof_stride < len) return ZK_EINVAL;
    for (uint32_t q = 0; q < B; q++) memcpy(proofs_out + (size_t)q * proof_stride, trs[q]->out.data(), len);
    return ZK_OK;
}

ZK_API(zk_poly_upload_canonical, (zk_ctx* c, zk_poly h, const uint64_t* host_canonical, size_t n), (c, h, host_canonical, n)) {
    int rc = zk_poly_upload(c, h, host_canonical, n);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    auto it = c->polys.find(h);
    if (it == c->polys.end()) return ZK_EINVAL;
    if ((rc = ctx_bind(c))) return rc;
    launch_to_mont(it->second.ptr, (uint32_t)n, c->stream);
    if (aud_sync(c, c->stream) != hipSuccess) return ZK_EHIP;
    return ZK_OK;
}
