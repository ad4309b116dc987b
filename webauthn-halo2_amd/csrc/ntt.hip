// ntt.hip — Stockham NTT over BN254 Fr for gfx950: register radix-4 butterflies on the carry-free field.
//
// Device replacement for halo2_proofs `arithmetic::best_fft` and the
// `EvaluationDomain::{lagrange_to_coeff, coeff_to_extended, extended_to_coeff}`
// wrappers around it (reference call sites
// halo2-circuits/src/ecc/ecdsa_p256.rs:366-373,416-423).  Natural order in,
// natural order out: out[i] = sum_j in[j] * w^(i*j).
//
// One launch = one Stockham pass of radix R = 2^log_r.  A 128-thread workgroup owns T consecutive
// butterfly groups j and the R x T tile (512 elements) lives in LDS.  What makes a pass cheap — the
// passes are bound by instruction issue, not by HBM — is that everything between the
// global load and the global store happens on the carry-free 9 x 29-bit form of field29.hip.h:
//   * a butterfly is ONE product (258 instructions) plus limb-wise add / subtract (9 instructions each,
//     no reduction): decimation in TIME, (u, v) -> (u + w v, u - w v), so that both outputs are sums of a
//     lazily bounded value and a fresh product (< 2p) and the bound grows by +3 per stage instead of doubling;
//   * two stages at a time run in registers (radix 4: four elements per lane, three twiddles), so a radix-2^7
//     pass makes four LDS round trips (4 + 4 + 4 + 2-point rounds) instead of seven; limbs are
//     re-normalised (carry propagation, no reduction) once per round, when they are written back;
//   * values are only reduced where they leave the pass: to "< 2^256" (a multiple of p subtracted by one
//     small multiply-add pass) between passes, to the canonical standard form with the one product that
//     also applies 1/N and the coset un-scaling on the last pass.
// Field forms: the engine's memory image is the Rust one (x * 2^256 mod p, canonical); inside a transform
// values are x * 2^261 mod p ("internal", field29.hip.h): the first pass converts on load (for free: the
// limbs of 32 x, or with the coset pre-scaling product, whose constant carries the factor), intermediate
// buffers hold internal values < 2^256 in eight words, the twiddle table handed to this file is in internal
// canonical form (w^i * 2^261), and the last pass's final product divides by 32 again.
// Global traffic per pass is one read + one write of the vector; coset scaling (zeta^i pre-multiply,
// zero-extension) is fused into the first pass's load and 1/N, coset un-scaling and truncation into the last
// pass's store.  blockIdx.y = vector: the coefficient / coset forms of a whole column batch take one launch per pass.
#include <stdlib.h>
#include <string.h>

#include "engine.h"
#include "field29.hip.h"

namespace zk {

#ifndef ZK_NTT_TILE_LOG  // build-time tuning knobs (tools/ab_variants.sh)
#define ZK_NTT_TILE_LOG 9
#endif
#ifndef ZK_NTT_MINW
#define ZK_NTT_MINW 1
#endif
#ifndef ZK_NTT_PAD
#define ZK_NTT_PAD 1
#endif
// The tile (elements a workgroup holds in LDS) is a template parameter of the pass kernel since round 6: 2^9 (512 elements x 36 B =
// 18 KiB of LDS, 128 lanes: ~7 workgroups share a CU), 2^10 (256 lanes, ~48 KiB with the in-tile twiddles) and 2^11 (512 lanes, ~96 KiB:
// one workgroup per CU — gfx950 gives a workgroup up to 160 KiB).  The larger tiles
// exist for TWO-PASS plans of the mid sizes (ntt_run): a pass costs a load / inter-pass twiddle product / reduce / store round
// per element whatever its radix, and radix 2^9 / 2^10 passes need tiles of at least two 32-byte columns to keep the global
// accesses 64 bytes wide.  ZK_NTT_TILE_LOG (build-time) pins one tile for every size (the round-3 tuning variants).
static constexpr int NTT_TILE_LOG = ZK_NTT_TILE_LOG;      // the default tile
static constexpr int NTT_TILE_LOG_MAX = 11;
typedef Fe29<FrParams> Fr29;

struct NttPassArgs {
    const Fr* in[NTT_MAX_BATCH];   // one vector per blockIdx.y
    Fr* out[NTT_MAX_BATCH];
    const Fr* tw;        // w_N^i * 2^261 mod p (canonical words), i in [0, N)
    const Fr* tw_last;   // FOLD passes: c * w_N^i * 2^256 (standard form, c = 1 or 1/N): the inter-pass twiddle of the LAST pass
                         // also carries the conversion internal -> standard (and 1/N), so that the store needs no product
    uint32_t log_n;
    uint32_t log_r;      // this pass's radix
    uint32_t log_ns;     // product of radices of earlier passes (0: first pass, input in standard form)
    uint32_t log_t;      // j's per workgroup
    uint32_t inverse;    // use w^-1 (index N - e)
    uint32_t last;       // last pass: output in canonical standard form
    uint32_t n_in;       // elements >= n_in read as zero (first pass only; else N)
    uint32_t n_out;      // elements >= n_out are not stored (last pass only; else N)
    uint32_t has_pre;    // multiply in[i] by pre[i % 3] on load (first pass)
    uint32_t has_post;   // multiply out[i] by post[i % 3] on store (last pass)
    Fr pre[3];           // pre-scale factors * 2^266 (plain words): product with a standard-form input is internal
    const Fr* pre_tab[NTT_MAX_BATCH];  // first pass, per vector (nullptr: none): in[i] is multiplied by pre_tab[i] (times 2^266, plain
                                       // words) instead of pre[i % 3] — the per-element twist of a coset transform (engine.hip coset3)
    Fr post[3];          // post-scale factors in standard Montgomery form: product with an internal value is standard
};

// ---- 9 x 29-bit element <-> LDS.  Tile elements are stored limb-major ("SoA": word = limb * tile + slot) with the slot
// index swizzled (bits 4-5 ^= bits 6-7) so that the access patterns of all rounds — lanes walk consecutive slots in runs
// of 4 .. 64, the runs 64 slots or more apart — spread over the 64 banks; the in-tile twiddles are 9 consecutive words.
#ifndef ZK_NTT_SOA
#define ZK_NTT_SOA 0
#endif
__device__ __forceinline__ uint32_t swz(uint32_t i) { return i ^ (((i >> 6) & 3u) << 4); }
__device__ __forceinline__ Fr29 tile_load(const uint32_t* lds, uint32_t tile, uint32_t row, uint32_t p, uint32_t t, uint32_t log_t) {
    Fr29 r;
#if ZK_NTT_SOA
    const uint32_t* q = lds + swz((p << log_t) | t);
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = q[i * tile];
#else
    const uint32_t* q = lds + p * row + t * 9;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = q[i];
#endif
    return r;
}
__device__ __forceinline__ void tile_store(uint32_t* lds, uint32_t tile, uint32_t row, uint32_t p, uint32_t t, uint32_t log_t, const Fr29& a) {
#if ZK_NTT_SOA
    uint32_t* q = lds + swz((p << log_t) | t);
#pragma unroll
    for (int i = 0; i < 9; i++) q[i * tile] = a.l[i];
#else
    uint32_t* q = lds + p * row + t * 9;
#pragma unroll
    for (int i = 0; i < 9; i++) q[i] = a.l[i];
#endif
}
__device__ __forceinline__ Fr29 lds_load29(const uint32_t* p) {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = p[i];
    return r;
}
__device__ __forceinline__ void lds_store29(uint32_t* p, const Fr29& a) {
#pragma unroll
    for (int i = 0; i < 9; i++) p[i] = a.l[i];
}

// a normalised (limbs < 2^29, top limb free), value < 160 p  ->  the same residue as an integer < 2^256:
// with t = floor(a / 2^254) subtract m p, m = floor(1.3125 t) (p / 2^254 = 0.7561, so 0 <= a - m p < 2.7 * 2^254)
__device__ __forceinline__ Fr29 reduce_below_2_256(const Fr29& a) {
    const uint32_t t = a.l[8] >> 22;
    const uint32_t m = (t * 84u) >> 6;
    Fr29 r;
    int64_t acc = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        acc += (int64_t)a.l[i] - (int64_t)((uint64_t)m * Lim29<FrParams>::P[i]);
        r.l[i] = (uint32_t)acc & M29;
        acc >>= 29;
    }
    acc += (int64_t)a.l[8] - (int64_t)((uint64_t)m * Lim29<FrParams>::P[8]);
    r.l[8] = (uint32_t)acc;
    return r;
}

// a normalised (limbs < 2^29), value < 160 p  ->  the canonical representative as 8 words.  q = floor(a / p) is estimated
// from the top limb: T = floor(a / 2^232) < 2^29.4, P8 = floor(p / 2^232) (22 bits); qe = floor(T / (P8 + 1)) is q or q - 1
// (T / (P8 + 1) <= a / p < (T + 1) / P8, the two differ by < 2^-13), taken as a multiplication by M = floor(2^53 / (P8 + 1))
// (error < T / 2^53): 0 <= a - qe p < 2 p, then one conditional subtraction
__device__ __forceinline__ Fr reduce_canonical(const Fr29& a) {
    constexpr uint32_t P8 = Lim29<FrParams>::P[8];
    constexpr uint64_t M = ((uint64_t)1 << 53) / (P8 + 1);  // < 2^32
    static_assert(M < ((uint64_t)1 << 32), "M must fit a word");
    const uint32_t qe = (uint32_t)(((uint64_t)a.l[8] * (uint32_t)M) >> 53);
    Fr29 r;
    int64_t acc = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        acc += (int64_t)a.l[i] - (int64_t)((uint64_t)qe * Lim29<FrParams>::P[i]);
        r.l[i] = (uint32_t)acc & M29;
        acc >>= 29;
    }
    acc += (int64_t)a.l[8] - (int64_t)((uint64_t)qe * Lim29<FrParams>::P[8]);
    r.l[8] = (uint32_t)acc;
    Fr o = from29(r);
    reduce_once(o);
    return o;
}

// decimation-in-time butterflies.  Bounds (value as a multiple of p ; limb bits), "fresh" = product output (< 2p ; 29):
//   bfly_mul    t = v w fresh;  u' = u + t (k_u + 2),  v' = u - t + 3p (k_u + 3)
//   bfly_plain  trivial twiddle, v bounded by KV:  u' = u + v,  v' = u - v + (KV + 1) p
template <int E = 29>
__device__ __forceinline__ void bfly_mul(Fr29& u, Fr29& v, const Fr29& w) {
    const Fr29 t = mul29(v, w);
    v = sub29<3, 29>(u, t);
    u = add29(u, t);
}
template <uint32_t K, int E>
__device__ __forceinline__ void bfly_plain(Fr29& u, Fr29& v) {
    const Fr29 t = v;
    v = sub29<K, E>(u, t);
    u = add29(u, t);
}

// first two stages (half = 1, 2) of an R-point DIT transform: the only non-trivial twiddle is the 4th root of unity.
// KIN bounds the loaded values (first pass: limbs of 32 x -> 32 p;  later passes: < 2^256 -> 5.3 p).
template <uint32_t KIN>
__device__ __forceinline__ void round0(Fr29& e0, Fr29& e1, Fr29& e2, Fr29& e3, const Fr29& w4) {
    bfly_plain<KIN + 1, 29>(e0, e1);          // e0 <= 2 KIN (limbs < 2^30), e1 <= 2 KIN + 1
    bfly_plain<KIN + 1, 29>(e2, e3);
    bfly_plain<2 * KIN + 1, 30>(e0, e2);      // sums of sums: limbs < 2^30 -> spread with E = 30;  <= 4 KIN + 1
    bfly_mul(e1, e3, w4);                     // <= 2 KIN + 4
}

// MODE 0: the general pass.
// MODE 1 (FOLD): the last pass of a transform of two or more passes whose output scaling is uniform (none, or 1/N): every
//   loaded element is multiplied by tw_last[.] = c w^e 2^256 — the inter-pass twiddle, the conversion internal -> standard
//   and c in ONE product (the trivial twiddles w^0 included: tw_last[0] = c 2^256) — and the store only reduces to the
//   canonical representative (reduce_canonical: ~85 instructions instead of a product + conversion, ~300).
// MODE 2 (ZQ): the first pass of a transform whose input is at least three quarters zeros (coeff_to_extended: n_in <= N/4):
//   only r < R/4 is loaded, into a compact array; positions brev(r) = 0 mod 4 are the only non-zero ones, so stages 0 and 1
//   (groups of four adjacent positions) just copy x to all four — round 0 is skipped, round 1 reads the compact array.
enum { NTT_GENERAL = 0, NTT_FOLD = 1, NTT_ZQ = 2 };
template <uint32_t KIN, int MODE, int TL = NTT_TILE_LOG>
#ifndef ZK_NTT_WAVES
#define ZK_NTT_WAVES 0
#endif
#if ZK_NTT_WAVES
__attribute__((amdgpu_waves_per_eu(ZK_NTT_WAVES, ZK_NTT_WAVES)))
#endif
__global__ __launch_bounds__(1 << (TL - 2), ZK_NTT_MINW) void ntt_pass_kernel(const NttPassArgs a) {
    constexpr uint32_t NTT_THREADS = 1u << (TL - 2);  // one radix-4 butterfly group per lane and round
    constexpr int NTT_TILE_LOG = TL;                  // (shadows the file-level default inside the kernel)
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const Fr* __restrict__ vin = a.in[blockIdx.y];
    Fr* __restrict__ vout = a.out[blockIdx.y];

    const uint32_t N = 1u << a.log_n;
    const uint32_t R = 1u << a.log_r;
    const uint32_t T = 1u << a.log_t;
    const uint32_t tile = R * T;
    const uint32_t j0 = blockIdx.x * T;
    const uint32_t ns_mask = (1u << a.log_ns) - 1;
    const uint32_t col_stride = N >> a.log_r;               // N / R
    const uint32_t tw_shift = a.log_n - a.log_ns - a.log_r;  // N / (Ns * R)
    const uint32_t nmask = N - 1;
    const uint32_t row = T * 9 + ZK_NTT_PAD;                 // words per position p (padded)
    uint32_t* wr = lds + (ZK_NTT_SOA ? 9 * tile : R * row);  // R / 2 in-tile twiddles w_R^j, 9 words each
    const uint32_t brev_shift = 32 - a.log_r;

    // ---- the R/2 twiddles of the in-tile stages (w_R^j = w_N^(j N/R), internal form) go to LDS once per workgroup
    for (uint32_t j = threadIdx.x; j < (R >> 1); j += NTT_THREADS) {
        const uint32_t ex = j << (a.log_n - a.log_r);
        lds_store29(wr + j * 9, to29(fe_load(a.tw + (a.inverse ? ((N - ex) & nmask) : ex))));
    }

    // ---- load: position brev(r) of column t  <-  in[j + r N/R] * pre * w_{Ns R}^{r (j mod Ns)}   (internal form)
    auto src_index = [&](uint32_t e, uint32_t& t, uint32_t& r) {
        t = e & (T - 1);
        r = e >> a.log_t;
        return j0 + t + r * col_stride;
    };
    auto tw_index = [&](uint32_t t, uint32_t r) {
        const uint32_t ex = (r * ((j0 + t) & ns_mask)) << tw_shift;
        return a.inverse ? ((N - ex) & nmask) : ex;  // 0 <=> trivial twiddle
    };
    const Fr* __restrict__ twl = MODE == NTT_FOLD ? a.tw_last : a.tw;  // inter-pass twiddles of this pass
    const Fr* __restrict__ ptab = (MODE == NTT_GENERAL && a.log_ns == 0) ? a.pre_tab[blockIdx.y] : nullptr;  // per-element pre-scale
    auto place = [&](uint32_t t, uint32_t r, uint32_t idx, const Fr& v, const Fr& w, uint32_t ti) {
        Fr29 x;
        if (idx < a.n_in) {
            if (MODE == NTT_FOLD) {
                x = mul29(to29(v), to29(w));  // every element, w^0 too: the product also converts (and scales)
            } else if (a.log_ns == 0) {
                // first pass: standard form in.  32 v is a valid internal form (bound 32 p); the coset factor's
                // constant carries 2^266 so that the product lands in internal form
                if (ptab) {
                    x = mul29(to29(v), to29(w));  // w = pre_tab[idx]
                } else {
                    const uint32_t m = a.has_pre ? idx % 3 : 0;
                    x = m ? mul29(to29(v), to29(a.pre[m])) : to29_x32(v);
                }
            } else {
                x = to29(v);  // internal value < 2^256
                if (ti) x = mul29(x, to29(w));
            }
        } else {
#pragma unroll
            for (int i = 0; i < 9; i++) x.l[i] = 0;
        }
        const uint32_t pos = a.log_r ? (__brev(r) >> brev_shift) : 0;
        tile_store(lds, tile, row, pos, t, a.log_t, x);
    };
    constexpr uint32_t EPT = (1u << NTT_TILE_LOG) / NTT_THREADS;  // elements per lane of a full tile
    uint32_t* cq = wr + (R >> 1) * 9;  // ZQ: the compact array [R / 4][T] of the non-zero quarter (same row pitch as the tile)
    if (MODE == NTT_ZQ) {
        // (host side: full tile, log_r >= 4, n_in <= N / 4, first pass)  one element per lane: r < R / 4 only
        const uint32_t e = threadIdx.x;
        const uint32_t t = e & (T - 1), r = e >> a.log_t;
        const uint32_t idx = j0 + t + r * col_stride;
        Fr29 x;
        if (idx < a.n_in) {
            const Fr v = fe_load(vin + idx);
            const uint32_t m = a.has_pre ? idx % 3 : 0;
            x = m ? mul29(to29(v), to29(a.pre[m])) : to29_x32(v);
        } else {
#pragma unroll
            for (int i = 0; i < 9; i++) x.l[i] = 0;
        }
        tile_store(cq, tile >> 2, row, __brev(r) >> (brev_shift + 2), t, a.log_t, x);
    } else if (tile == (1u << NTT_TILE_LOG)) {
        // full tile: the lane's loads (data and inter-pass twiddles) are all issued before the first product
        Fr v[EPT], w[EPT];
        uint32_t tt[EPT], rr[EPT], ii[EPT], ti[EPT];
#pragma unroll
        for (uint32_t k = 0; k < EPT; k++) {
            ii[k] = src_index(threadIdx.x + k * NTT_THREADS, tt[k], rr[k]);
            ti[k] = a.log_ns ? tw_index(tt[k], rr[k]) : 0;
            v[k] = ii[k] < a.n_in ? fe_load(vin + ii[k]) : Fr::zero();
            if (ptab) w[k] = ii[k] < a.n_in ? fe_load(ptab + ii[k]) : Fr::zero();
            else w[k] = (ti[k] || MODE == NTT_FOLD) ? fe_load(twl + ti[k]) : Fr::zero();
        }
#pragma unroll
        for (uint32_t k = 0; k < EPT; k++) place(tt[k], rr[k], ii[k], v[k], w[k], ti[k]);
    } else {
        for (uint32_t e = threadIdx.x; e < tile; e += NTT_THREADS) {
            uint32_t t, r;
            const uint32_t idx = src_index(e, t, r);
            const uint32_t ti = a.log_ns ? tw_index(t, r) : 0;
            const Fr v = idx < a.n_in ? fe_load(vin + idx) : Fr::zero();
            const Fr w = ptab ? (idx < a.n_in ? fe_load(ptab + idx) : Fr::zero()) : (ti || MODE == NTT_FOLD) ? fe_load(twl + ti) : Fr::zero();
            place(t, r, idx, v, w, ti);
        }
    }
    __syncthreads();

    // ---- log_r DIT stages over the position index, two per round in registers
#ifndef ZK_NTT_EXP_NOLDS
#define ZK_NTT_EXP_NOLDS 0   // 1: timing experiment — values stay in registers between rounds (WRONG results)
#endif
    Fr29 keep[4];
    (void)keep;
    uint32_t s = 0;
    if (MODE == NTT_ZQ) {
        s = 2;  // stages 0 and 1 copy the one non-zero element of every group of four: nothing to compute
    } else if (a.log_r >= 2) {
        // stages 0 and 1: groups of four adjacent positions
        const Fr29 w4 = lds_load29(wr + (R >> 2) * 9);  // w_R^(R/4)
        for (uint32_t q = threadIdx.x; q < (tile >> 2); q += NTT_THREADS) {
            const uint32_t t = q & (T - 1), g = q >> a.log_t;
            const uint32_t b = g << 2;
            Fr29 e0 = tile_load(lds, tile, row, b, t, a.log_t), e1 = tile_load(lds, tile, row, b + 1, t, a.log_t),
                 e2 = tile_load(lds, tile, row, b + 2, t, a.log_t), e3 = tile_load(lds, tile, row, b + 3, t, a.log_t);
            round0<KIN>(e0, e1, e2, e3, w4);
#if ZK_NTT_EXP_NOLDS
            keep[0] = norm29(e0); keep[1] = norm29(e1); keep[2] = norm29(e2); keep[3] = norm29(e3);
#else
            tile_store(lds, tile, row, b, t, a.log_t, norm29(e0));
            tile_store(lds, tile, row, b + 1, t, a.log_t, norm29(e1));
            tile_store(lds, tile, row, b + 2, t, a.log_t, norm29(e2));
            tile_store(lds, tile, row, b + 3, t, a.log_t, norm29(e3));
#endif
        }
#if !ZK_NTT_EXP_NOLDS
        __syncthreads();
#endif
        s = 2;
    }
    for (; s + 1 < a.log_r; s += 2) {
        const uint32_t h = 1u << s;
        for (uint32_t q = threadIdx.x; q < (tile >> 2); q += NTT_THREADS) {
            const uint32_t t = q & (T - 1), g = q >> a.log_t;
            const uint32_t lo = g & (h - 1);
            const uint32_t base = ((g >> s) << (s + 2)) | lo;
#if ZK_NTT_EXP_NOLDS
            Fr29 e0 = keep[0], e1 = keep[1], e2 = keep[2], e3 = keep[3];
#else
            Fr29 e0, e1, e2, e3;
            if (MODE == NTT_ZQ && s == 2) {  // positions base + 4 i all hold the compact element (base >> 2) + i
                const uint32_t cb = (g >> 2) << 2;
                e0 = tile_load(cq, tile >> 2, row, cb, t, a.log_t), e1 = tile_load(cq, tile >> 2, row, cb + 1, t, a.log_t);
                e2 = tile_load(cq, tile >> 2, row, cb + 2, t, a.log_t), e3 = tile_load(cq, tile >> 2, row, cb + 3, t, a.log_t);
            } else {
                e0 = tile_load(lds, tile, row, base, t, a.log_t), e1 = tile_load(lds, tile, row, base + h, t, a.log_t);
                e2 = tile_load(lds, tile, row, base + 2 * h, t, a.log_t), e3 = tile_load(lds, tile, row, base + 3 * h, t, a.log_t);
            }
#endif
            {
                const Fr29 w1 = lds_load29(wr + (lo << (a.log_r - s - 1)) * 9);   // w_{2h}^lo
                bfly_mul(e0, e1, w1);
                bfly_mul(e2, e3, w1);
            }
            bfly_mul(e0, e2, lds_load29(wr + (lo << (a.log_r - s - 2)) * 9));        // w_{4h}^lo
            bfly_mul(e1, e3, lds_load29(wr + ((lo + h) << (a.log_r - s - 2)) * 9));  // w_{4h}^(lo + h)
#if ZK_NTT_EXP_NOLDS
            keep[0] = norm29(e0); keep[1] = norm29(e1); keep[2] = norm29(e2); keep[3] = norm29(e3);
            if (s + 3 >= a.log_r) {
#endif
            tile_store(lds, tile, row, base, t, a.log_t, norm29(e0));
            tile_store(lds, tile, row, base + h, t, a.log_t, norm29(e1));
            tile_store(lds, tile, row, base + 2 * h, t, a.log_t, norm29(e2));
            tile_store(lds, tile, row, base + 3 * h, t, a.log_t, norm29(e3));
#if ZK_NTT_EXP_NOLDS
            }
#endif
        }
#if ZK_NTT_EXP_NOLDS
        if (s + 3 >= a.log_r)
#endif
        __syncthreads();
    }
    if (s < a.log_r) {  // one stage left (odd log_r, or log_r == 1): half = R / 2
        const uint32_t h = R >> 1;
        for (uint32_t q = threadIdx.x; q < (tile >> 1); q += NTT_THREADS) {
            const uint32_t t = q & (T - 1), lo = q >> a.log_t;
            Fr29 e0 = tile_load(lds, tile, row, lo, t, a.log_t), e1 = tile_load(lds, tile, row, lo + h, t, a.log_t);
            if (a.log_r == 1) {
                bfly_plain<KIN + 1, 29>(e0, e1);
            } else {
                bfly_mul(e0, e1, lds_load29(wr + lo * 9));  // w_R^lo
            }
            tile_store(lds, tile, row, lo, t, a.log_t, norm29(e0));
            tile_store(lds, tile, row, lo + h, t, a.log_t, norm29(e1));
        }
        __syncthreads();
    }

    // ---- store: out[(j / Ns) Ns R + (j mod Ns) + r Ns] = X_r
    const Fr29 one_std = const_pow2_29<256, FrParams>();  // internal -> standard: times 2^256 * 2^-261
    for (uint32_t e = threadIdx.x; e < tile; e += NTT_THREADS) {
        uint32_t t, r;
        if (a.log_ns == 0) {  // dst = j*R + r : contiguous over r
            r = e & (R - 1);
            t = e >> a.log_r;
        } else {              // contiguous over j within an Ns block
            t = e & (T - 1);
            r = e >> a.log_t;
        }
        const uint32_t j = j0 + t;
        const uint32_t dst = ((j >> a.log_ns) << (a.log_ns + a.log_r)) + (j & ns_mask) + (r << a.log_ns);
        if (dst < a.n_out) {
            const Fr29 x = tile_load(lds, tile, row, r, t, a.log_t);
            Fr o;
            if (MODE == NTT_FOLD) {
                o = reduce_canonical(x);
            } else if (a.last) {
                o = from29(mul29(x, a.has_post ? to29(a.post[dst % 3]) : one_std));
                reduce_once(o);
            } else {
                o = from29(reduce_below_2_256(x));
            }
            fe_store(vout + dst, o);
        }
    }
}

// Twiddle table: tw[i] = w^i, i in [0, N).  Each thread seeds w^(i0) by
// square-and-multiply, then walks 64 consecutive powers.  `scale` multiplies every entry (standard form): the
// NTT's own table is made with scale = 32, i.e. w^i * 2^261 as plain words.
__global__ void ntt_twiddle_kernel(Fr* tw, Fr w, Fr scale, uint32_t n) {
    const uint32_t CH = 64;
    const uint32_t i0 = (blockIdx.x * blockDim.x + threadIdx.x) * CH;
    if (i0 >= n) return;
    Fr cur = scale;
    Fr base = w;
    for (uint32_t e = i0; e; e >>= 1) {
        if (e & 1) cur = fe_mul(cur, base);
        base = fe_sqr(base);
    }
    for (uint32_t k = 0; k < CH && i0 + k < n; k++) {
        fe_store(tw + i0 + k, cur);
        cur = fe_mul(cur, w);
    }
}

void launch_twiddles(Fr* tw, const Fr& w, uint32_t n, hipStream_t st) {
    const uint32_t threads = (n + 63) / 64;
    hipLaunchKernelGGL(ntt_twiddle_kernel, dim3((threads + 63) / 64), dim3(64), 0, st, tw, w, Fr::one(), n);
}

void launch_twiddles_scaled(Fr* tw, const Fr& w, const Fr& scale, uint32_t n, hipStream_t st) {
    const uint32_t threads = (n + 63) / 64;
    hipLaunchKernelGGL(ntt_twiddle_kernel, dim3((threads + 63) / 64), dim3(64), 0, st, tw, w, scale, n);
}

static Fr fr_small_mont(uint32_t x) {
    Fr a = Fr::zero();
    a.v[0] = x;
    return fe_to_mont(a);
}

void launch_twiddles_internal(Fr* tw, const Fr& w, uint32_t n, hipStream_t st) {
    const uint32_t threads = (n + 63) / 64;
    hipLaunchKernelGGL(ntt_twiddle_kernel, dim3((threads + 63) / 64), dim3(64), 0, st, tw, w, fr_small_mont(32), n);
}

#ifndef ZK_NTT_FOLD  // build-time switches for A/B runs (tools/ab_variants.sh)
#define ZK_NTT_FOLD 1
#endif
#ifndef ZK_NTT_ZQ
#define ZK_NTT_ZQ 1
#endif
static constexpr bool FOLD_ON = ZK_NTT_FOLD != 0, ZQ_ON = ZK_NTT_ZQ != 0;

// Plan the passes of a 2^log_n transform: radices as even as possible, each <= max_log_r.
int ntt_plan(uint32_t log_n, uint32_t max_log_r, uint32_t bits[8]) {
    if (log_n == 0) return 0;
    const uint32_t np = (log_n + max_log_r - 1) / max_log_r;
    uint32_t rem = log_n;
    for (uint32_t p = 0; p < np; p++) {
        bits[p] = (rem + (np - p) - 1) / (np - p);
        rem -= bits[p];
    }
    return (int)np;
}

// Runs all passes.  `a` is the input (left intact unless it is also `b`/`c`);
// ping-pongs between b and c so that the result lands in `dst`.  dst must be
// distinct from src unless a scratch `tmp` (N elements) is supplied.
hipError_t ntt_run(const NttJob& job, hipStream_t st) {
    const uint32_t log_n = job.log_n;
    const uint32_t N = 1u << log_n;
    const uint32_t batch = job.batch ? job.batch : 1;
    if (batch > NTT_MAX_BATCH) return hipErrorInvalidValue;
    const Fr* srcs[NTT_MAX_BATCH];
    Fr* dsts[NTT_MAX_BATCH];
    for (uint32_t b = 0; b < batch; b++) {
        srcs[b] = job.batch ? job.srcs[b] : job.src;
        dsts[b] = job.batch ? job.dsts[b] : job.dst;
    }
    uint32_t bits[8];
    // The plan: largest radix and tile by transform size.  Measured (tools/ntt_sweep.py, one vector, ms; 3 passes of 2^7 on the 2^9
    // tile -> the plan below): 2^15 0.044 -> 0.032, 2^16 0.047 -> 0.034 (two passes of 2^8, tile 2^9); 2^17 0.052 -> 0.043, 2^18 0.076 ->
    // 0.051 (two passes of 2^9, tile 2^10); 2^19 0.103 -> 0.079, 2^20 0.174 -> 0.158 (two passes of 2^10, tile 2^11); 2^21 stays at
    // three passes of 2^7 (two passes of 2^11 / 2^10 on the 2^11 tile: 0.37 against 0.30 ms — one 32-byte column per tile row).
    // An explicit radix (ZK_OPT_NTT_MAX_RADIX_LOG2) takes the smallest tile that holds it; a build that pins ZK_NTT_TILE_LOG keeps
    // its tile and the old default radix for every size.
    uint32_t tl = NTT_TILE_LOG, max_r = 7;
    if (job.max_log_r) {
        max_r = job.max_log_r;
        if (ZK_NTT_TILE_LOG == 9) tl = max_r < 9 ? 9 : max_r > (uint32_t)NTT_TILE_LOG_MAX ? NTT_TILE_LOG_MAX : max_r;
    } else if (ZK_NTT_TILE_LOG == 9) {
        if (log_n <= 16) {
            max_r = 8;
        } else if (log_n <= 18) {
            max_r = 9;
            tl = 10;
        } else if (log_n <= 20) {
            max_r = 10;
            tl = 11;
        }
    }
    if (max_r > tl) max_r = tl;
    const int np = ntt_plan(log_n, max_r, bits);
    if (np == 0) {  // N == 1
        for (uint32_t b = 0; b < batch; b++)
            if (dsts[b] != srcs[b]) {
                hipError_t e = hipMemcpyAsync(dsts[b], srcs[b], sizeof(Fr), hipMemcpyDeviceToDevice, st);
                if (e != hipSuccess) return e;
            }
        return hipSuccess;
    }
    // Ping-pong so that the last pass writes dst and no pass runs in place:
    // pass p writes bufs[(np - 1 - p) & 1] with bufs = {dst, tmp}.
    bool inplace = false;
    for (uint32_t b = 0; b < batch; b++) inplace = inplace || srcs[b] == dsts[b];
    if (job.tmp == nullptr && (np > 1 || inplace)) return hipErrorInvalidValue;
    const Fr* cur_in[NTT_MAX_BATCH];
    for (uint32_t b = 0; b < batch; b++) {
        cur_in[b] = srcs[b];
        if (srcs[b] == dsts[b] && (np & 1)) {
            // first pass would read and write dst: stage the input in tmp
            hipError_t e = hipMemcpyAsync(job.tmp + (size_t)b * N, srcs[b], sizeof(Fr) * (size_t)(job.n_in < N ? job.n_in : N),
                                          hipMemcpyDeviceToDevice, st);
            if (e != hipSuccess) return e;
            cur_in[b] = job.tmp + (size_t)b * N;
        }
    }
    // pre-scale constants with the 2^266 that turns (standard input) x (constant) into internal form:
    // c * 2^256 (Montgomery image) times 2^10
    Fr pre266[3];
    for (int i = 0; i < 3; i++) pre266[i] = job.has_pre ? fe_mul(job.pre[i], fr_small_mont(1024)) : Fr::zero();
    int which = (np & 1) ? 0 : 1;  // 0: dst, 1: tmp
    uint32_t log_ns = 0;
    for (int p = 0; p < np; p++) {
        NttPassArgs a;
        memset(&a, 0, sizeof(a));
        for (uint32_t b = 0; b < batch; b++) {
            a.in[b] = cur_in[b];
            a.out[b] = which ? job.tmp + (size_t)b * N : dsts[b];
            a.pre_tab[b] = (p == 0 && job.batch) ? job.pre_tabs[b] : nullptr;
        }
        a.tw = job.tw;
        a.tw_last = job.tw_last;
        a.log_n = log_n;
        a.log_r = bits[p];
        a.log_ns = log_ns;
        uint32_t log_t = tl - a.log_r;
        if (log_t > log_n - a.log_r) log_t = log_n - a.log_r;
        a.log_t = log_t;
        a.inverse = job.inverse;
        a.last = p == np - 1;
        a.n_in = (p == 0) ? job.n_in : N;
        a.n_out = (p == np - 1) ? job.n_out : N;
        // the last of several passes takes its conversion (and a uniform output scaling) from tw_last: no final product
        const bool fold = FOLD_ON && np >= 2 && p == np - 1 && job.tw_last != nullptr && (!job.has_post || job.tw_last_has_post);
        // the first pass of a transform over an input that is >= 3/4 zeros skips its first two stages
        const bool zq = ZQ_ON && p == 0 && (uint64_t)job.n_in * 4 <= N && a.log_r >= 4 && a.log_r + log_t == tl;
        a.has_pre = (p == 0) ? job.has_pre : 0;
        a.has_post = (p == np - 1 && !fold) ? job.has_post : 0;
        for (int i = 0; i < 3; i++) {
            a.pre[i] = pre266[i];
            a.post[i] = job.post[i];
        }
        const uint32_t blocks = N >> (a.log_r + a.log_t);
        const size_t row_bytes = ZK_NTT_SOA ? ((size_t)36 << a.log_t) : (((size_t)9 << a.log_t) + ZK_NTT_PAD) * 4;
        const size_t lds = (row_bytes << a.log_r) + ((size_t)36 << a.log_r) / 2 + (zq ? (row_bytes << (a.log_r - 2)) : 0);
        const int mode = zq ? 0 : p == 0 ? 1 : fold ? 2 : 3;
#define ZK_NTT_LAUNCH(TLV)                                                                                                             \
    do {                                                                                                                               \
        const dim3 grid(blocks, batch), blk(1u << ((TLV) - 2));                                                                        \
        if (mode == 0) hipLaunchKernelGGL((ntt_pass_kernel<32, NTT_ZQ, TLV>), grid, blk, lds, st, a);                                  \
        else if (mode == 1) hipLaunchKernelGGL((ntt_pass_kernel<32, NTT_GENERAL, TLV>), grid, blk, lds, st, a);                        \
        else if (mode == 2) hipLaunchKernelGGL((ntt_pass_kernel<6, NTT_FOLD, TLV>), grid, blk, lds, st, a);                            \
        else hipLaunchKernelGGL((ntt_pass_kernel<6, NTT_GENERAL, TLV>), grid, blk, lds, st, a);                                        \
    } while (0)
#if ZK_NTT_TILE_LOG == 9
        if (tl == 11) ZK_NTT_LAUNCH(11);
        else if (tl == 10) ZK_NTT_LAUNCH(10);
        else ZK_NTT_LAUNCH(9);
#else
        ZK_NTT_LAUNCH(ZK_NTT_TILE_LOG);
#endif
#undef ZK_NTT_LAUNCH
        for (uint32_t b = 0; b < batch; b++) cur_in[b] = a.out[b];
        which ^= 1;
        log_ns += bits[p];
    }
    return hipGetLastError();
}

}  // namespace zk
