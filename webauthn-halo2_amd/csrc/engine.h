// engine.h — internal declarations shared by the HIP translation units of
// libzkmi355.so (not part of the public C-ABI; that is include/zkmi355.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "ec.hip.h"
#include "field.hip.h"

namespace zk {

// ---- NTT (ntt.hip) ---------------------------------------------------------
static constexpr uint32_t NTT_MAX_BATCH = 96;  // vectors transformed by one launch (grid.y): 32 columns x the three cosets of poly.hip "three cosets"
struct NttJob {
    const Fr* src;      // n_in elements (rest of the 2^log_n vector is zero)
    Fr* dst;            // 2^log_n elements (n_out stored)
    Fr* tmp;            // 2^log_n scratch per vector (required when more than one pass, or src == dst)
    // batch > 1: the same transform over srcs[b] -> dsts[b] (src / dst unused); tmp holds batch x 2^log_n
    uint32_t batch;
    const Fr* srcs[NTT_MAX_BATCH];
    Fr* dsts[NTT_MAX_BATCH];
    const Fr* tw;       // twiddle table in the NTT's internal form: w^i * 2^261 mod p, i < 2^log_n (launch_twiddles_internal)
    // optional: c * w^i in the STANDARD form (c = 1, or post[0] when tw_last_has_post): lets the last pass of a multi-pass
    // transform fold the final conversion (and a uniform output scaling) into its inter-pass twiddles (ntt.hip NTT_FOLD)
    const Fr* tw_last;
    uint32_t tw_last_has_post;
    uint32_t log_n;
    uint32_t inverse;   // use w^-1
    uint32_t n_in, n_out;
    uint32_t has_pre, has_post;
    Fr pre[3], post[3]; // period-3 scale factors by index (coset zeta powers, 1/N)
    // batch mode, optional per vector: element i of srcs[b] is multiplied by pre_tabs[b][i] (times 2^266 as plain words, like
    // `pre`) on the way in instead of pre[i % 3]: the twist of a transform over the coset (zeta w_4n^j) H (engine.hip, coset3)
    const Fr* pre_tabs[NTT_MAX_BATCH];
    uint32_t max_log_r; // 0 = default
};
hipError_t ntt_run(const NttJob& job, hipStream_t st);
void launch_twiddles(Fr* tw, const Fr& w, uint32_t n, hipStream_t st);           // w^i, standard Montgomery form
void launch_twiddles_scaled(Fr* tw, const Fr& w, const Fr& scale, uint32_t n, hipStream_t st);  // scale * w^i, standard form
void launch_twiddles_internal(Fr* tw, const Fr& w, uint32_t n, hipStream_t st);  // w^i * 2^261 (plain words): ntt.hip's own form
int ntt_plan(uint32_t log_n, uint32_t max_log_r, uint32_t bits[8]);

// ---- the three-coset route of a quotient with three pieces (poly.hip "three cosets") ----
struct Coset3Consts {  // constants of the 3 x 3 solve, standard form
    Fr inv2, inv_2z, zi, inv_2z2;  // 1/2, 1/(2 z), z i, 1/(2 z^2) with z = zeta^n, i = w_4n^n
    Fr zinv[3];                    // zeta^-(m mod 3)
};
void launch_coset3_pre(const Fr* tw_ext, uint32_t n, uint32_t j, const Fr zp1024[3], Fr* tab, hipStream_t st);
void launch_coset3_relayout(const Fr* const* src, Fr* const* dst, uint32_t count, uint32_t n, hipStream_t st);
void launch_coset3_combine(Fr* h, const Fr* tw_ext, uint32_t n, const Coset3Consts& k, hipStream_t st);

// ---- MSM (msm.hip) ---------------------------------------------------------
struct MsmWorkspace;  // opaque, sized for a maximum n and a maximum number of columns per launch
static constexpr uint32_t MSM_MAX_BATCH = 256;
MsmWorkspace* msm_workspace_create(size_t max_n, uint32_t c, hipError_t* err, uint32_t max_batch = 1);
void msm_workspace_destroy(MsmWorkspace* ws);
uint32_t msm_auto_window(size_t n, uint32_t override_c = 0);  // fixed-base mode (the resident SRS's window tables)
uint32_t msm_auto_window_generic(size_t n);                   // arbitrary bases
uint32_t msm_num_windows(uint32_t c);
size_t msm_ws_max_n(const MsmWorkspace* ws);
uint32_t msm_ws_max_batch(const MsmWorkspace* ws);
uint32_t msm_ws_window(const MsmWorkspace* ws);
void msm_ws_set_t1_mode(MsmWorkspace* ws, uint32_t mode);  // ZK_OPT_MSM_T1: 1 one lane per bucket, 0 / 2 parts + segmented tree (default)
bool msm_ws_last_pass_wide(const MsmWorkspace* ws);
// table[w * n + i] = 2^(c w) * bases[i] (affine), w < msm_num_windows(c)
// (the wide path's tables — 15 / 16-bit windows, msm_table_is_internal — hold the points in the accumulation's internal form,
// x * 2^261: they are read by the fixed-base MSM only)
hipError_t msm_build_table(const G1Affine* bases, uint32_t n, uint32_t c, G1Affine* table, hipStream_t st);
bool msm_table_is_internal(uint32_t c, size_t n);
// Launches the whole device pipeline on `st`; the bit sums (XYZZ) are copied to `host_window_sums`
// asynchronously.  `table` != nullptr selects the fixed-base mode: `batch` scalar vectors (columns) against
// the same bases in ONE pass, one bucket set per column; *nwin_out = batch independent results, each finished
// with msm_finish_host(sums + q * stride, 1, c).  Without a table batch must be 1 and *nwin_out windows need
// the host Horner: msm_finish_host(sums, *nwin_out, c).  A workspace of 15 / 16-bit windows whose table indexes fit 24 bits runs the
// fixed-base mode on the "wide path" (msm.hip: dense entry lists, restartable lanes, row / column tail): results through
// msm_ws_finish_fixed, msm_ws_sums_per_result sums each.
hipError_t msm_run(MsmWorkspace* ws, const Fr* const* scalars_list, uint32_t batch, const G1Affine* bases, size_t n,
                   hipStream_t st, G1X* host_window_sums, uint32_t* nwin_out, uint32_t* c_out,
                   hipEvent_t* accum_events = nullptr, const G1Affine* table = nullptr, uint32_t table_stride = 0,
                   hipStream_t tail_st = nullptr, hipEvent_t head_done = nullptr, bool bases_may_be_identity = true);
// whether any of the n points is the identity (synchronises `st`; d_word / h_word: one device word and one pinned host word of
// scratch): a basis without one takes the unchecked accumulation loop
hipError_t msm_bases_have_identity(const G1Affine* bases, uint32_t n, hipStream_t st, uint32_t* d_word, uint32_t* h_word, bool* out);
// G1X entries per result in host_window_sums (fixed-base mode)
uint32_t msm_sums_per_result(uint32_t c);
// fixed-base mode, by workspace (the wide path of 15 / 16-bit windows hands over 16 sums per column)
uint32_t msm_ws_sums_per_result(const MsmWorkspace* ws);
G1Jac msm_ws_finish_fixed(const MsmWorkspace* ws, const G1X* sums);
// wide path: lanes of the unchecked accumulation that must be redone with the checked loop (0 unless the basis is degenerate),
// as reported with the pass's sums, and that redo + the tail again (results into host_window_sums once `st` has drained)
uint32_t msm_wide_redo_count(const MsmWorkspace* ws, const G1X* host_window_sums, uint32_t batch);
hipError_t msm_wide_redo(MsmWorkspace* ws, uint32_t batch, size_t n, hipStream_t st, G1X* host_window_sums, const G1Affine* table);
// Host-side finish: Horner over windows -> Jacobian (Montgomery).
G1Jac msm_finish_host(const G1X* window_sums, uint32_t nwin, uint32_t c);

// ---- host helpers (hostmath.cpp) --------------------------------------------
G1Affine g1_jac_to_affine_host(const G1Jac& p);

}  // namespace zk
