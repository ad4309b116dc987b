// prover.h — internal declarations of the device prover (prover_kernels.hip,
// quotient.hip, prover.hip).  Not part of the public C-ABI.
#pragma once
#include "engine.h"

namespace zk {

static constexpr uint32_t MAX_LC = 40;       // inputs of one linear combination
static constexpr uint32_t MAX_CHUNK = 3;     // permutation columns per grand product (degree - 2)
// sized for every row of the reference's bench_ecdsa.config, down to k=11 (291 gate + 53 lookup columns,
// 4 constants columns); the argument blocks below live in device memory, not in the kernarg segment
static constexpr uint32_t MAX_ADV = 352;
static constexpr uint32_t MAX_FIX = 304;
static constexpr uint32_t MAX_PERM = 352;
static constexpr uint32_t MAX_CHUNKS = 176;
static constexpr uint32_t MAX_LOOKUPS = 56;
static constexpr uint32_t MAX_TERMS = MAX_ADV + 2 * MAX_CHUNKS + 1 + 5 * MAX_LOOKUPS;  // y-combination terms of the quotient
static_assert(MAX_CHUNKS + MAX_LOOKUPS <= 256, "gp_chain_kernel scans all grand products of a proof in one 256-lane block");
static constexpr uint32_t NO_SELECTOR = 0xffffffffu;  // gate column whose never-enabled selector owns no fixed column (it shares another gate's: Layout::gate_sel)
static constexpr uint32_t BLINDING_FACTORS = 6;  // max(3, 4 queries per gate column) + 2

struct LincombArgs {
    Fr* out;
    uint32_t n, count, accumulate, sub0;
    Fr sub0_val;  // subtracted from coefficient 0
    uint32_t sub_low_n, pad_[3];
    Fr sub_low[8];  // subtracted from coefficients 0 .. sub_low_n - 1 (a low-degree remainder polynomial)
    const Fr* in[MAX_LC];
    uint32_t len[MAX_LC];
    uint32_t unit[MAX_LC];  // coefficient is 1
    Fr c[MAX_LC];
};

// one input of a long linear combination whose argument list lives in device memory (launch_lincomb_terms)
struct LcTerm {
    const Fr* in;   // n coefficients
    uint64_t pad_;
    Fr c;
};
struct LcLow {
    Fr v[8];
};

struct ChaChaKey {
    uint32_t w[8];
};

struct LookupScratch {
    uint32_t *hist, *present, *absent, *off, *dex, *aex;  // lookup 0's arrays, T + 1 entries each
    uint32_t* bsum;    // 3 x ceil(T / 1024) block sums
    uint32_t* err;     // one flag for all lookups
    uint32_t stride;   // words between the arrays of consecutive lookups
};
struct LkPtrs {
    const Fr* inp[MAX_LOOKUPS];
    Fr* ap[MAX_LOOKUPS];
    Fr* sp[MAX_LOOKUPS];
};

struct PermArgs {
    uint32_t n, ncols;
    const Fr* values[MAX_CHUNK];
    const Fr* sigma[MAX_CHUNK];
    Fr delta[MAX_CHUNK];  // delta^(global column index)
    const Fr* tw;         // w^i
    Fr beta, gamma;
    Fr *num, *den;
};

// one thread per row of the extended coset (expressions pinned by
// reference proving-server/P256Verifier.yul:406-552)
struct QuotientArgs {
    uint32_t log_ext, n_gate, n_adv, n_chunks, chunk_len, n_perm, n_lookups, single;
    int32_t last_rot;     // -(blinding_factors + 1)
    uint32_t fx_table, fx_qlookup;
    const Fr* adv[MAX_ADV];       // extended cosets
    const Fr* fix[MAX_FIX];
    uint32_t fx_sel[MAX_ADV];
    const Fr* sigma[MAX_PERM];
    const Fr* perm_val[MAX_PERM]; // coset of the column each permutation column refers to
    const Fr* z[MAX_CHUNKS];
    const Fr* lk_z[MAX_LOOKUPS];
    const Fr* lk_a[MAX_LOOKUPS];
    const Fr* lk_s[MAX_LOOKUPS];
    const Fr* lk_in[MAX_LOOKUPS]; // A >= 2: lookup advice coset; A == 1: unused (q_lookup * adv[0])
    const Fr *l0, *l_last, *l_active;
    const Fr* xs;                 // zeta * w_ext^i: the coset points
    Fr beta, gamma, delta;        // times 32 (the kernel's internal form, quotient.hip)
    Fr t_inv[4];                  // 1 / ((zeta w_ext^i)^n - 1), period 4, standard form; all 1 when divide == 0
    uint32_t divide, n_terms;     // divide: multiply by t_inv (divide_by_vanishing_poly); 0: the bare numerator of evaluate_h
    Fr* out;
    Fr delta_chunk[MAX_CHUNKS];   // 32 delta^(c chunk_len): where chunk c's column factors start (the sliced kernel)
    Fr ypow[MAX_TERMS];           // 32 y^(T-1-j) for term j of the y-combination, T = n_terms
};
// terms of the y-combination: gates, 2 + (chunks - 1) + chunks permutation terms, 5 per lookup
static constexpr uint32_t quotient_terms(uint32_t n_gate, uint32_t n_chunks, uint32_t n_lookups) {
    return n_gate + 2 + (n_chunks - 1) + n_chunks + 5 * n_lookups;
}

void launch_to_mont(Fr* a, uint32_t n, hipStream_t st);
void launch_mul(Fr* out, const Fr* a, const Fr* b, uint32_t n, hipStream_t st);
void launch_lincomb(const LincombArgs& a, hipStream_t st);
// out = sum_j terms[j].c * terms[j].in (- sub0_val on coefficient 0, - low[i] on the first low_n coefficients); `h_terms` is
// pinned host memory the caller keeps intact until the stream has passed this point, `d_terms` its place in device memory
void launch_lincomb_terms(LcTerm* h_terms, LcTerm* d_terms, uint32_t count, Fr* out, uint32_t n, bool sub0, const Fr& sub0_val,
                          const Fr* low, uint32_t low_n, hipStream_t st);
void launch_scale(Fr* a, const Fr& c, uint32_t n, hipStream_t st);
void launch_chacha_fr(const ChaChaKey& key, uint64_t start_block, Fr* out, uint32_t count, hipStream_t st);
void launch_scan_u32(const uint32_t* in, uint32_t* out, uint32_t m, hipStream_t st);
void launch_lookup_permute(const LkPtrs& ptrs, uint32_t count, uint32_t usable, uint32_t T, LookupScratch& s, hipStream_t st);
void launch_perm_numden(const PermArgs& a, hipStream_t st);
struct LkNumDenArgs {
    const Fr *ap, *sp, *inp, *tab;
    Fr *num, *den;
};
struct CopyPair {
    const Fr* src;
    Fr* dst;
};
// batched forms for the many-column shapes (one launch for all chunks / lookups / columns; argument blocks in device memory)
void launch_perm_numden_batch(const PermArgs* d_args, uint32_t count, uint32_t n, hipStream_t st);
void launch_lk_numden_batch(const LkNumDenArgs* d_args, uint32_t count, const Fr& beta, const Fr& gamma, uint32_t n, hipStream_t st);
void launch_copy_columns(const CopyPair* d_pairs, uint32_t count, uint32_t n, hipStream_t st);
void launch_lk_numden(const Fr* ap, const Fr* sp, const Fr* inp, const Fr* tab, const Fr& beta, const Fr& gamma, Fr* num,
                      Fr* den, uint32_t n, hipStream_t st);
void launch_frac(const Fr* num, const Fr* den, Fr* frac, uint32_t n, hipStream_t st);
void launch_prefix_product(const Fr* f, Fr* z, uint32_t n, const Fr* init_dev, const Fr& init_val, Fr* tmp_local,
                           Fr* tmp_tot, hipStream_t st);
// one grand product of a batch (launch_gp_batch_*): z[0] = init, z[i+1] = z[i] * num[i] / den[i]
struct GpItem {
    const Fr* num;
    const Fr* den;
    Fr* loc_p;    // n: inclusive prefix products of num within 2048-element blocks
    Fr* loc_r;    // n: inclusive suffix products of den within blocks
    Fr* tot_p;    // blocks: block totals, then exclusive offsets
    Fr* tot_r;
    Fr* z;
    uint32_t chain;  // init = the previous item's z at the chain row (permutation chunks); otherwise 1
    uint32_t pad_;
};
uint32_t gp_blocks(uint32_t n);
void launch_gp_batch_scan(const GpItem* d_items, uint32_t nprod, uint32_t n, Fr* q_dev, hipStream_t st);
void launch_gp_batch_apply(const GpItem* d_items, uint32_t nprod, uint32_t n, uint32_t chain_row, const Fr* q_inv_dev, Fr* k_dev,
                           Fr* init_dev, hipStream_t st);
// a short run of rows of one column (blinding rows), staged on the host
struct RowEntry {
    Fr vals[8];
    Fr* dst;
    uint32_t count;
    uint32_t pad_;
};
void launch_scatter_rows(const RowEntry* d_entries, uint32_t count, hipStream_t st);
static constexpr uint32_t KD_MAX_BATCH = 6;  // divisions per launch (SHPLONK / GWC have at most six rotation sets)
uint32_t kate_division_scratch(uint32_t n);  // elements of scratch per division
void launch_kate_division(const Fr* p, Fr* q, uint32_t n, const Fr& z, Fr* scratch, hipStream_t st);
void launch_kate_division_batch(const Fr* const* p, Fr* const* q, const Fr* z, uint32_t count, uint32_t n, Fr* scratch,
                                hipStream_t st);
void launch_quotient_dev(const QuotientArgs* d_args, uint32_t log_ext, uint32_t log_slices, hipStream_t st, bool cosets3 = false);
uint32_t quotient_log_slices(uint32_t log_ext, uint32_t n_gate);

// poly.hip
struct EvalItem {
    const Fr* poly;
    uint64_t pad_;
    Fr x;
    // filled by launch_eval_batch: x^(2^l), l < 8; y = x^256; z^(2^l), l < 9, with z = x^(coefficients per workgroup)
    Fr pw[8];
    Fr y;
    Fr zpw[9];
};
void launch_eval_batch(EvalItem* h_items, EvalItem* d_items, uint32_t count, uint32_t n, Fr* scratch, Fr* out,
                       hipStream_t st);
uint32_t eval_blocks(uint32_t n);
void launch_eval(const Fr* c, uint32_t n, const Fr& x, Fr* scratch, hipStream_t st);

}  // namespace zk
