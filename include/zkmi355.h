/*
 * zkmi355.h — C ABI of libzkmi355.so, the MI355X (gfx950) proving engine that
 * sits behind halo2_proofs::plonk::create_proof for the reference's
 * secp256r1-ECDSA circuit.
 *
 * Every entry point replaces a Rust routine of the (un-vendored) halo2_proofs /
 * halo2curves crates that the reference reaches from its three create_proof
 * call sites:
 *     halo2-circuits/src/ecc/ecdsa_p256.rs:366-373   (EvmTranscript + GWC, /prove_evm)
 *     halo2-circuits/src/ecc/ecdsa_p256.rs:416-423   (Blake2b + SHPLONK, /prove)
 *     halo2-circuits/src/ecc/ecdsa_p256.rs:555-562   (bench_secp256r1_ecdsa)
 * and from keygen at ecdsa_p256.rs:258-260.  The upstream routine replaced is
 * named on each declaration; INTEGRATION.md shows the Rust `extern "C"` shim.
 *
 * Conventions
 *   - plain pointers and sizes only; the caller owns every host buffer, the
 *     engine never keeps a host pointer past return;
 *   - memory images are those of the Rust types: Fr / Fq = 4 x u64 little-endian
 *     limbs in Montgomery form (R = 2^256); G1Affine = x || y (64 B), identity
 *     (0,0); G1 = Jacobian x || y || z (96 B), identity z = 0;
 *   - every function returns 0 on success or a negative ZK_E* code, never
 *     throws; outputs are untouched on error; no CPU fallback exists — a missing
 *     device is ZK_ENODEV;
 *   - a zk_ctx is bound to one HIP device and one stream and serialises its
 *     callers internally; use one context per worker thread / GPU;
 *   - the engine has no entropy source of its own: the seam and phase-level entry points are
 *     deterministic maps, and the one entry point that needs blinding values, zk_prove, expands the 32-byte seed the
 *     CALLER supplies with ChaCha20 (halo2's ChaCha20Rng stream and draw order) — the host draws that seed from its
 *     own RNG (the reference uses OsRng, ecdsa_p256.rs:362,412,550); a fixed or predictable seed forfeits zero-knowledge.
 */
#ifndef ZKMI355_H
#define ZKMI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZK_OK 0
#define ZK_EINVAL (-1)   /* bad size / argument / handle */
#define ZK_ENOMEM (-2)   /* device or host allocation failed */
#define ZK_EHIP (-3)     /* HIP runtime error (zk_last_hip_error) */
#define ZK_ENODEV (-4)   /* no usable gfx950 device */
#define ZK_ESTATE (-5)   /* missing prerequisite (SRS / key not loaded) */
#define ZK_EWITNESS (-6) /* witness violates the circuit (lookup input not in table): halo2's
                            Error::ConstraintSystemFailure */
#define ZK_EINTERNAL (-7) /* a C++ exception was stopped at the boundary (nothing is ever thrown across it), or — under ZK_OPT_STREAM_AUDIT — the
                            stream audit found an enqueue that is not ordered after the buffers it touches (zk_audit_report) */
#define ZK_ELAYOUT (-8)  /* zk_keygen / zk_pk_read: the selector columns do not fit the layout the key is built for — halo2's
                            compress_selectors would combine them differently (two used gate selectors that share no row, a used
                            selector that is never enabled, an "idle" one that is; or 2 * num_idle_gate_columns > num_advice:
                            the surplus never-enabled selectors would pair up in all-zero columns of their own): another vk
                            digest and other gates than the engine's closed form, so the key is refused rather than made to
                            prove something else */

typedef struct zk_ctx zk_ctx;
typedef uint64_t zk_poly; /* opaque device-resident vector of Fr; 0 is never valid */

#define ZK_BASIS_MONOMIAL 0 /* ParamsKZG::commit          (basis g)          */
#define ZK_BASIS_LAGRANGE 1 /* ParamsKZG::commit_lagrange (basis g_lagrange) */

/* ---- context ------------------------------------------------------------- */
int zk_device_count(void);
/* PCI address of a device ("0000:c1:00.0", cap >= 16): /sys/bus/pci/devices/<id>/{numa_node,local_cpulist} name its NUMA node —
 * an 8-GPU host binds each GPU's worker threads and staging buffers there (bench.py does) */
int zk_device_pci_bus_id(int device_id, char* out, size_t cap);
/* free and total device memory in bytes (hipMemGetInfo): a host sizes the number of resident proof pipelines by it (a k = 19
 * pipeline is ~ 5.3 GiB beside the first one's 6.4, docs in DESIGN.md section 2) */
int zk_device_mem_info(int device_id, size_t* free_bytes, size_t* total_bytes);
/* page-locked host memory for buffers handed to zk_poly_upload / zk_poly_upload_canonical (one DMA at the bus rate instead of a
 * staged copy out of pageable memory); NULL on failure */
void* zk_host_alloc(size_t bytes);
void zk_host_free(void* p);
int zk_ctx_create(int device_id, zk_ctx** out);
/* a further context on ctx's device that SHARES ctx's resident SRS (bases and window tables, read-only, as loaded at this
 * moment): one context per proof pipeline / host thread without a copy of the tables each.  Either context may later load
 * another SRS for itself; the shared memory is freed when its last user lets go (destroying `parent` first is fine). */
int zk_ctx_create_shared(zk_ctx* parent, zk_ctx** out);
void zk_ctx_destroy(zk_ctx* ctx);
const char* zk_strerror(int code);
int zk_last_hip_error(const zk_ctx* ctx); /* raw hipError_t of the last ZK_EHIP */
int zk_sync(zk_ctx* ctx);                 /* hipStreamSynchronize on the context stream */
/* tuning options (measurement tools, tests); value 0 restores the built-in choice.  The engine reads no
 * environment variables. */
#define ZK_OPT_MSM_WINDOW 1          /* signed window bits of the fixed-base MSM, 9..17; takes effect at the next SRS load */
#define ZK_OPT_MSM_BATCH 2           /* columns per fixed-base MSM pass, 1..256 */
#define ZK_OPT_NTT_MAX_RADIX_LOG2 3  /* largest radix of one NTT pass, 1..11 (clamped to the tile) */
#define ZK_OPT_GP_BATCH_INVERT 4     /* 1: grand products always take halo2's batch_invert form (the fallback path) */
#define ZK_OPT_MSM_TAIL_STREAM 5     /* where the MSM reduction tails run: 0 auto, 1 the context's side stream, 2 its main stream.
                                        Auto: the side stream while at most ZK_OPT_MSM_TAIL_MAIN_ABOVE contexts are ACTIVE on the
                                        device, the main stream beyond.  A context is active if it enqueued an MSM pass within the
                                        last 4 ms through ANY entry point (zk_prove, zk_commit, zk_commit_batch, zk_msm_srs,
                                        zk_msm_bn254, keygen): four host threads on the phase-level ABI are seen like four
                                        zk_prove calls.  The count is PROCESS-LOCAL — other processes on the same GPU are not seen:
                                        a host that runs several worker processes per device pins the regime with 1 / 2.
                                        Why it matters: the HIP runtime maps a process's streams onto GPU_MAX_HW_QUEUES hardware
                                        queues (4 unless that environment variable of the HIP runtime says otherwise) and streams
                                        that share a queue run in order — with more than four streams busy some pipeline's kernels
                                        wait behind another's accumulation (DESIGN.md "Streams and hardware queues"; measured: 4
                                        pipelines 103.7 proofs/s on one stream each against 99.1 with a side stream each) */
#define ZK_OPT_MSM_TAIL_MAIN_ABOVE 6 /* the auto threshold above, 1..64; 0 restores the measured default (2, for the runtime's default
                                        of four hardware queues: two contexts x two streams fill them) */
#define ZK_OPT_BATCH_PASS_COLUMNS 7  /* zk_prove_batch: columns per MSM pass, 1..256; 0 = max(min(2 x batch, 8), the single prover's pass width).
                                         The lanes' MSM workspaces grow to that width on their next pass and KEEP it for the life of the context */
#define ZK_OPT_XFORM_STREAM 8        /* zk_prove: where a proof's column transforms (values -> coefficients -> extended coset) run:
                                        0 auto (beside the MSM passes on a stream of their own while this is the only active context
                                        of the process on the device — a lone proof —, in order on the main stream otherwise: see
                                        ZK_OPT_MSM_TAIL_STREAM on hardware queues), 1 always the side stream, 2 always the main stream */
#define ZK_OPT_MSM_STREAM 9          /* zk_prove: where a proof's MSM passes (sort head + accumulation) run: 0 auto (a stream of their own
                                        for a lone proof — with ZK_OPT_XFORM_STREAM's auto rule: main, tail, transform and MSM stream are
                                        the runtime's four hardware queues —, so that the glue kernels of the next phase do not queue
                                        behind an accumulation), 1 always that stream, 2 always the main stream */
#define ZK_OPT_MSM_T1 10             /* first kernel of the MSM reduction tail (the sum of a bucket's partial sums): 0 / 2 parts of <= 8 partial
                                        sums + a segmented tree (default), 1 one lane per bucket (a quarter fewer instructions, a dependent chain
                                        twice as long: measured equal under load, slower for a lone proof).  Same bytes either way */
#define ZK_OPT_STREAM_AUDIT 11       /* debug: 1 switches on the happens-before ledger of the context's streams (csrc/audit.h): every enqueue
                                        names the buffers it reads and writes, and one whose stream is not ordered after the buffer's last
                                        writer (or, for a write, its last readers) makes the entry point return ZK_EINTERNAL instead of ZK_OK —
                                        the class of bug that yields wrong proof bytes once in a thousand runs shows on the first run that takes
                                        the faulty path.  zk_audit_report tells what was found.  0 (default) switches it off; 2 = the
                                        audit's self-test: on, AND zk_prove takes a knowingly unordered path (its column transforms alternate
                                        between two streams per call without ordering their shared scratch — round 5's first form): every
                                        proof whose transforms come in more than one call must then return ZK_EINTERNAL */
#define ZK_OPT_STREAM_PRIORITY 12    /* experiment: dispatch priority of the context's MAIN stream — 0 normal (default), 1 high, 2 low
                                        (hipStreamCreateWithPriority).  Set it right after zk_ctx_create, before any other call: the stream
                                        is made again.  Pipelines of one device at DIFFERENT priorities do not share the chip evenly, so
                                        their chip-filling kernels stop ending together (docs/experiments.md) */
#define ZK_OPT_QUOTIENT_DOMAIN 13    /* zk_prove, circuits whose quotient has THREE pieces (two or more advice columns: deg h < 3n): h can be
                                        taken over three of the four cosets of halo2's extended domain — three n-point transforms per column
                                        instead of one 4n-point one, 3n quotient rows, a 3 x 3 solve per coefficient: the same pieces, the same
                                        proof bytes, a quarter of that work less.  0 (default) = auto: that route for columns of 2^16 rows or
                                        more (the proving server's k = 17 among them; the many-column rows below lose by it), 1 = always the
                                        whole extended domain (round 5's route), 2 = three cosets wherever h has three pieces.
                                        zk_prove_batch follows the same rule; the phase-level entry points (zk_coeff_to_extended, zk_quotient,
                                        zk_extended_to_coeff) always work on the whole domain in halo2's order */
#define ZK_OPT_ACTIVITY_HOLD 14      /* how a context inside zk_prove / zk_prove_batch counts for the automatic stream rules of the OTHER contexts
                                        of its device (ZK_OPT_MSM_TAIL_STREAM, ZK_OPT_XFORM_STREAM): 0 (default) active for the whole call — its
                                        quotient / evaluation / multi-open phases enqueue no MSM pass for longer than the 4 ms window under load,
                                        and the count dipped to two or three several times per proof; 1 = by its stamps alone (round 5's rule);
                                        2 = this context counts as active from now on, whatever entry points it uses, until the option is set
                                        to 0 or 1 — for the worker contexts of a host that drives the phase-level ABI (its calls between two MSM
                                        passes are invisible to the 4 ms window) */
int zk_ctx_set_option(zk_ctx* ctx, int option, int64_t value);

/* ---- fine-grained drop-in seam (host buffers in, host buffers out) --------
 * replaces halo2_proofs::arithmetic::best_multiexp(coeffs: &[Fr], bases: &[G1Affine]) -> G1 */
int zk_msm_bn254(zk_ctx* ctx, const uint64_t* scalars_mont /* n x 4 */,
                 const uint64_t* bases_affine_mont /* n x 8 */, size_t n,
                 uint64_t out_jacobian_mont[12]);
/* the same product over the RESIDENT SRS: replaces the body of ParamsKZG::commit (basis g: best_multiexp(&coeffs,
 * &self.g[..n])) / ParamsKZG::commit_lagrange (basis g_lagrange), where the Rust host knows by construction which
 * basis it multiplies against.  Only the n <= 2^k scalars are uploaded; the MSM runs on the window tables built at
 * zk_srs_setup / zk_srs_load / zk_srs_read.  (zk_msm_bn254 never guesses: it always uploads its bases.) */
int zk_msm_srs(zk_ctx* ctx, int basis /* ZK_BASIS_* */, const uint64_t* scalars_mont /* n x 4 */, size_t n,
               uint64_t out_jacobian_mont[12]);
/* replaces halo2_proofs::arithmetic::best_fft(a: &mut [Fr], omega: Fr, log_n: u32); in place,
 * natural order in and out */
int zk_ntt_bn254_fr(zk_ctx* ctx, uint64_t* a_mont /* 2^log_n x 4 */, const uint64_t omega_mont[4],
                    uint32_t log_n);

/* ---- SRS (ParamsKZG<Bn256>) ----------------------------------------------
 * replaces ParamsKZG::setup(k, ChaCha20Rng::from_seed(seed)) as run by halo2-base gen_srs(k)
 * (ecdsa_p256.rs:258,338,388): s = Fr::from_u512(first 64 keystream bytes);
 * g[i] = [s^i]G1, g_lagrange[i] = [L_i(s)]G1, both generated on the device. */
int zk_srs_setup(zk_ctx* ctx, uint32_t k, const uint8_t seed[32]);
/* adopt caller-supplied bases (the in-memory ParamsKZG of a Rust host: params.g / params.g_lagrange, n = 2^k points
 * each, affine Montgomery).  The arrays are copied; nothing is remembered about them.  Loading an SRS invalidates
 * every proving key made under the previous one (zk_prove / zk_vk_export return ZK_ESTATE for them). */
int zk_srs_load(zk_ctx* ctx, uint32_t k, const uint64_t* g, const uint64_t* g_lagrange);
int zk_srs_export(zk_ctx* ctx, int basis, uint64_t* out_affine_mont /* n x 8 */, size_t first, size_t count);
int zk_srs_k(const zk_ctx* ctx); /* -1 if none */
/* how zk_commit runs over the resident SRS: signed window width and the number of windows (= bucket
 * additions per scalar; best_multiexp's `c` and `segments`, halo2_proofs arithmetic.rs [RECALLED]).
 * window_bits == 0: no window-multiple tables (k < 10).  For measurement (bench.py's ALU roofline). */
int zk_srs_msm_plan(const zk_ctx* ctx, uint32_t* window_bits, uint32_t* windows);

/* ---- the reference's files (SRS, proving key, verifying key) -------------------------------------------------
 * halo2_proofs `SerdeFormat` (helpers.rs): how field elements and points are laid out in a file.  The reference writes
 * and reads its keys with RawBytes (ecdsa_p256.rs:261-270, 339-343, 389-393); `Params::write` is RawBytes too. */
#define ZK_SERDE_PROCESSED 0           /* canonical little-endian field elements, compressed points */
#define ZK_SERDE_RAW_BYTES 1           /* in-memory Montgomery limbs, validated on read */
#define ZK_SERDE_RAW_BYTES_UNCHECKED 2 /* the same bytes, no validation on read */
/* replaces ParamsKZG::write_custom (halo2-base gen_srs writes ./params/kzg_bn254_{k}.srs): u32 LE k | g | g_lagrange | g2 |
 * s_g2.  out == NULL: only *len is set.  ZK_ESTATE after zk_srs_load until zk_srs_set_g2 supplies the G2 half. */
int zk_srs_write(zk_ctx* ctx, int format, uint8_t* out, size_t cap, size_t* len);
/* replaces ParamsKZG::read_custom (run by gen_srs on EVERY request in the reference, ecdsa_p256.rs:338,388): decodes
 * (decompresses / validates on the device), builds the window tables, keeps the SRS resident.  ZK_EINVAL on a
 * malformed file; the previously loaded SRS is gone in that case. */
int zk_srs_read(zk_ctx* ctx, const uint8_t* bytes, size_t len, int format);
/* G2 half of a ParamsKZG adopted with zk_srs_load (G2Affine memory images: x.c0 || x.c1 || y.c0 || y.c1, Montgomery) */
int zk_srs_set_g2(zk_ctx* ctx, const uint64_t g2[16], const uint64_t s_g2[16]);

/* ---- resident polynomials -------------------------------------------------- */
int zk_poly_alloc(zk_ctx* ctx, size_t n, zk_poly* out);
/* the handle dies; the memory is parked in the context (a few vectors, at most 2 GiB) for the next zk_poly_alloc of the same
 * length, because hipFree waits for the whole device — a host that allocates and frees around every request would otherwise
 * stall every other context of the GPU once per request */
int zk_poly_free(zk_ctx* ctx, zk_poly p);
/* Hand-over of a resident vector between two contexts of one device, no copy and no cross-context lock: the owner detaches
 * (its stream is drained first; the handle dies, a process-wide token is returned), the new owner attaches (new handle).  A
 * loader context with its own stream and host thread uploads the next request's columns while the proving context is inside
 * zk_prove.  A detached vector belongs to nobody until it is attached (ZK_EINVAL: unknown token, or another device). */
int zk_poly_detach(zk_ctx* ctx, zk_poly p, uint64_t* token);
int zk_poly_attach(zk_ctx* ctx, uint64_t token, zk_poly* out);
/* Frees a detached vector that will never be attached (the loader failed between staging and adoption, the target context
 * is gone): a token that is neither attached nor discarded keeps its device memory for the life of the process. */
int zk_poly_discard(uint64_t token);
int zk_poly_len(zk_ctx* ctx, zk_poly p, size_t* out);
int zk_poly_upload(zk_ctx* ctx, zk_poly p, const uint64_t* host_mont, size_t n);
int zk_poly_download(zk_ctx* ctx, zk_poly p, uint64_t* host_mont, size_t n);
int zk_poly_copy(zk_ctx* ctx, zk_poly dst, zk_poly src);
/* rows [first, first + count) from the host: the blinding rows a host appends to a column the device made (halo2's provers push
 * `blinding_factors` random rows onto a', s' and every z before committing), without shipping the column */
int zk_poly_upload_range(zk_ctx* ctx, zk_poly p, size_t first, const uint64_t* host_mont, size_t count);
/* dst[dst_first ..] = src[src_first .. src_first + count) (the h pieces: n-coefficient slices of the quotient) */
int zk_poly_copy_range(zk_ctx* ctx, zk_poly dst, size_t dst_first, zk_poly src, size_t src_first, size_t count);
/* out = sum_j coeffs[j] * in[j] - (sub_low[0] + sub_low[1] X + .. + sub_low[n_low - 1] X^(n_low - 1)), n_low <= 8 (0: nothing
 * subtracted); all vectors of one length, out none of the inputs: the linear combinations of the multi-open provers — GWC's
 * sum_i v^i (p_i(X) - e_i) (n_low = 1), SHPLONK's sum_j y^j (P_j(X) - R_j(X)) with the remainders R_j of degree < |rotation set| —
 * and h(X) = sum_i x^(n i) h_i(X) */
int zk_poly_lincomb(zk_ctx* ctx, zk_poly out, const zk_poly* in, const uint64_t* coeffs_mont /* count x 4 */, size_t count,
                    const uint64_t* sub_low_mont /* n_low x 4 */, size_t n_low);

/* replaces ParamsKZG::commit / commit_lagrange (MSM against the resident SRS) + to_affine */
int zk_commit(zk_ctx* ctx, zk_poly p, int basis, uint64_t out_affine_mont[8]);
/* the same for `count` polynomials of one length: the columns share MSM passes (one bucket set per column, one
 * accumulation launch for several columns) — how create_proof commits its advice columns, (a', s'), grand
 * products and quotient pieces.  out: count x 8 limbs. */
int zk_commit_batch(zk_ctx* ctx, const zk_poly* polys, size_t count, int basis, uint64_t* out_affine_mont);
/* replaces EvaluationDomain::lagrange_to_coeff (in place: iNTT, x 1/n) */
int zk_lagrange_to_coeff(zk_ctx* ctx, zk_poly p);
/* replaces EvaluationDomain::coeff_to_lagrange (in place) */
int zk_coeff_to_lagrange(zk_ctx* ctx, zk_poly p);
/* replaces EvaluationDomain::coeff_to_extended: dst (2^ext_k) = NTT of zero-extended src scaled by zeta^i */
int zk_coeff_to_extended(zk_ctx* ctx, zk_poly src, zk_poly dst_ext);
/* replaces EvaluationDomain::extended_to_coeff: in place on a 2^ext_k vector; first n_out coefficients valid */
int zk_extended_to_coeff(zk_ctx* ctx, zk_poly ext, size_t n_out);
/* replaces arithmetic::eval_polynomial(poly, x) */
int zk_eval(zk_ctx* ctx, zk_poly p, const uint64_t x_mont[4], uint64_t out_mont[4]);
/* replaces arithmetic::kate_division(p, z): q = (p - p(z)) / (X - z).  q has p's length (its top coefficient is 0: the
 * quotient is one coefficient shorter); q may be p (in place).  The multi-open provers (GWC / SHPLONK) divide with it. */
int zk_kate_division(zk_ctx* ctx, zk_poly p, const uint64_t z_mont[4], zk_poly q);

/* ---- keygen / create_proof ---------------------------------------------------
 * The config row that selects the column shape: reference CircuitParams
 * (halo2-circuits/src/ecc/ecdsa_p256.rs:44-55; rows in src/configs/bench_ecdsa.config). */
typedef struct {
    uint32_t k;                 /* "degree" */
    uint32_t num_advice;
    uint32_t num_lookup_advice;
    uint32_t num_fixed;
    uint32_t lookup_bits;
    uint32_t num_idle_gate_columns; /* trailing gate columns whose selector is never enabled (at most half of num_advice):
                                       halo2's selector compression puts the t-th such selector into the fixed column of
                                       gate t — no column of its own — and replaces the pair by q (2 - q) / q (1 - q) */
} zk_circuit_params;
typedef uint64_t zk_pk; /* opaque: proving key + verifying key + prover workspace, device resident */

#define ZK_TRANSCRIPT_BLAKE2B 0 /* Blake2bWrite<_, G1Affine, Challenge255<_>>  (ecdsa_p256.rs:415) */
#define ZK_TRANSCRIPT_EVM 1     /* snark-verifier EvmTranscript (Keccak-256)   (ecdsa_p256.rs:365) */
#define ZK_SCHEME_DEFAULT 0     /* SHPLONK for Blake2b, GWC for EVM — the reference's pairings */
#define ZK_SCHEME_GWC 1         /* ProverGWC     (ecdsa_p256.rs:368) */
#define ZK_SCHEME_SHPLONK 2     /* ProverSHPLONK (ecdsa_p256.rs:418) */

/* replaces keygen_vk + keygen_pk (ecdsa_p256.rs:259-260) for a synthesized circuit: `fixed_canonical`
 * holds the fixed columns (n_fix x n x 4 limbs, canonical integers, column order: constants, range
 * table, selectors); `copies` the copy constraints as (perm_col_a, row_a, perm_col_b, row_b) with
 * permutation columns ordered [constants..., gate advice..., lookup advice...].  Needs the SRS of k. */
int zk_keygen(zk_ctx* ctx, const zk_circuit_params* params, const uint64_t* fixed_canonical /* n_fixed_columns x n x 4 */,
              size_t n_fixed_columns /* must equal the shape's fixed-column count: ZK_EINVAL otherwise */,
              const uint32_t* copies, size_t n_copies, zk_pk* out);
/* the vk digest every transcript starts with (halo2 `vk.transcript_repr`: Blake2b-512 of the pinned vk's Debug rendering,
 * reduced mod r): zk_keygen / zk_pk_read compute halo2's own value (csrc/vkrepr.h; pinned by the reference's k = 17 literal,
 * proving-server/P256Verifier.yul:34) for every shape, never-enabled gate columns included (bench_ecdsa.config rows k <= 13:
 * their combined selectors are rendered as compress_selectors builds them; no known answer exists for those rows).  A host
 * that knows better sets its value here (Montgomery image). */
int zk_pk_set_transcript_repr(zk_ctx* ctx, zk_pk pk, const uint64_t transcript_repr_mont[4]);
int zk_pk_free(zk_ctx* ctx, zk_pk pk);
/* the VerifyingKey half: commitments (affine Montgomery) and transcript_repr; counts = {n_fixed, n_perm} */
int zk_vk_export(zk_ctx* ctx, zk_pk pk, uint64_t* fixed_commitments, uint64_t* perm_commitments,
                 uint64_t transcript_repr[4], uint32_t counts[2]);
/* replaces VerifyingKey::write (ecdsa_p256.rs:266-270): u32 BE k | u32 BE #fixed | fixed commitments | permutation
 * commitments | selector bits.  out == NULL: only *len is set. */
int zk_vk_write(zk_ctx* ctx, zk_pk pk, int format, uint8_t* out, size_t cap, size_t* len);
/* adopts the Rust host's VerifyingKey (a VerifyingKey::write image) for a resident key: ZK_EINVAL unless its commitments
 * and selectors are the key's own; `transcript_repr` (may be NULL) is then what every transcript starts with. */
int zk_vk_load(zk_ctx* ctx, zk_pk pk, const uint8_t* vk_bytes, size_t len, int format, const uint64_t transcript_repr_mont[4]);
/* replaces ProvingKey::write (ecdsa_p256.rs:261-265): vk | l0 | l_last | l_active_row | fixed values / polys / cosets |
 * permutation values / polys / cosets (k=17: 336 MiB, k=19: 768 MiB in RawBytes).  out == NULL: only *len is set. */
int zk_pk_write(zk_ctx* ctx, zk_pk pk, int format, uint8_t* out, size_t cap, size_t* len);
/* replaces ProvingKey::read::<_, ECDSACircuit<Fr>> (ecdsa_p256.rs:339-343, 389-393 — on every request there, once
 * here): the key material comes from the file image, the column shape from `params` (what `ECDSACircuit::configure`
 * tells halo2), transcript_repr from the caller (NULL: computed as zk_keygen does).  Needs the SRS of params->k. */
int zk_pk_read(zk_ctx* ctx, const zk_circuit_params* params, const uint8_t* bytes, size_t len, int format,
               const uint64_t transcript_repr_mont[4], zk_pk* out);
/* the column shape of a key, for a host that drives the phases itself:
 * out = {k, extended k, #advice columns, #fixed columns, #permutation columns, #permutation chunks, #lookups, #h pieces} */
int zk_pk_shape(zk_ctx* ctx, zk_pk pk, uint32_t out[8]);
/* replaces plonk::evaluation::Evaluator::evaluate_h — and, with divide != 0, the EvaluationDomain::divide_by_vanishing_poly
 * that follows it in create_proof — for a Rust host that keeps halo2's own prover flow and off-loads phase by phase (the
 * [patch] route of INTEGRATION.md).  Every operand is a resident vector over the extended coset (2^(k+2) elements, as
 * zk_coeff_to_extended makes them): the advice columns, the permutation grand products z (one per chunk), and per lookup
 * the triple (permuted input a', permuted table s', product zL) — lookup_ext holds 3 * n_lookups handles in that order.
 * The fixed / sigma / l_0 / l_last / l_active cosets are the key's own.  beta, gamma, y: Montgomery images (theta does not
 * enter: every lookup of this circuit family is a single expression).  out_ext receives h on the extended coset
 * (zk_extended_to_coeff then gives the h pieces); it must not alias an input. */
int zk_quotient(zk_ctx* ctx, zk_pk pk, const zk_poly* advice_ext, size_t n_advice, const zk_poly* perm_z_ext, size_t n_chunks,
                const zk_poly* lookup_ext, size_t n_lookups, const uint64_t beta[4], const uint64_t gamma[4], const uint64_t y[4],
                int divide, zk_poly out_ext);
/* ---- the provers between the commitments, for a host that drives the phases itself (its transcript, its RNG, its blinding:
 * examples/prove_host_phases.cpp builds a whole proof from these and the calls above) ------------------------------------------
 * replaces plonk::lookup::prover's `permute_expression_pair` for every lookup of the key (halo2_proofs
 * plonk/lookup/prover.rs, reached from create_proof, ecdsa_p256.rs:366-373): from the advice columns (Lagrange values) the
 * permuted input a' and permuted table s' of each lookup, rows 0 .. n - 8 (the usable rows; the host writes the 7 rows behind
 * them — its blinding — with zk_poly_upload_range).  ZK_EWITNESS: an input is not in the table. */
int zk_lookup_permute(zk_ctx* ctx, zk_pk pk, const zk_poly* advice, size_t n_advice, zk_poly* permuted_input /* out, n_lookups */,
                      zk_poly* permuted_table /* out, n_lookups */, size_t n_lookups);
/* replaces lookup::prover `commit_product`: the grand product zL of every lookup (rows 0 .. n - 7; the host blinds the last 6) */
int zk_lookup_product(zk_ctx* ctx, zk_pk pk, const zk_poly* advice, size_t n_advice, const zk_poly* permuted_input,
                      const zk_poly* permuted_table, size_t n_lookups, const uint64_t beta[4], const uint64_t gamma[4],
                      zk_poly* z_out /* n_lookups */);
/* replaces plonk::permutation::prover `commit`: the grand products z of the permutation argument, one per chunk of columns
 * (zk_pk_shape), chunk c starting from chunk c - 1's value at the last usable row (rows 0 .. n - 7; the host blinds the rest) */
int zk_permutation_product(zk_ctx* ctx, zk_pk pk, const zk_poly* advice, size_t n_advice, const uint64_t beta[4],
                           const uint64_t gamma[4], zk_poly* z_out /* n_chunks */, size_t n_chunks);
/* a copy of one of the key's own polynomials (coefficient form, n coefficients) into the caller's vector: the fixed columns
 * (index in QUERY order: constants, lookup table, selector columns — the order of the fixed evaluations in a proof) and the
 * permutation polynomials sigma (permutation-column order) that create_proof evaluates at x and opens (`pk.fixed_polys`,
 * `pk.permutation.polys` of halo2's ProvingKey) */
#define ZK_PK_FIXED_POLY 0
#define ZK_PK_SIGMA_POLY 1
int zk_pk_export_poly(zk_ctx* ctx, zk_pk pk, int which, size_t index, zk_poly dst);
/* replaces the n `Fr::random(&mut rng)` draws of plonk::vanishing::prover `commit` (the random polynomial): coefficient i =
 * the Fr::random of ChaCha20 block first_block + i under `chacha_key` — what rand_chacha's ChaCha20Rng::from_seed(key) yields
 * for its draws first_block .. first_block + n - 1 when every draw is an Fr::random (as in create_proof).  The RNG stays the
 * host's: it hands over key and position and skips n draws; a host with another RNG uploads the column. */
int zk_random_poly(zk_ctx* ctx, const uint8_t chacha_key[32], uint64_t first_block, zk_poly out);
/* bytes zk_prove will write for this key / transcript / scheme (what `transcript.finalize().len()` is in
 * the reference, e.g. 960 at k=19 Blake2b, halo2-circuits/src/results/ecdsa_bench.csv:2) */
int zk_proof_size(zk_ctx* ctx, zk_pk pk, int transcript, int scheme, size_t* out);
/* replaces plonk::create_proof (ecdsa_p256.rs:366-373, 416-423, 555-562) for one circuit with no
 * instances.  `advice` are resident columns (Lagrange values, Montgomery, n rows each; the last 7 rows
 * are overwritten by blinding in a private copy).  Randomness: ChaCha20Rng::from_seed(rng_seed), one
 * 64-byte block per Fr::random in halo2's draw order.  Returns the proof bytes the transcript writer
 * would hold after `finalize()`.  proof_out == NULL: only *proof_len is set. */
int zk_prove(zk_ctx* ctx, zk_pk pk, const zk_poly* advice, size_t n_advice, const uint8_t rng_seed[32],
             int transcript, int scheme, uint8_t* proof_out, size_t proof_cap, size_t* proof_len);
/* `batch` independent create_proof calls for ONE key in lock-step on this context: the reference's concurrent requests
 * (one Rocket worker thread per request, proving-server/src/main.rs:457-472, each inside create_proof, ecdsa_p256.rs:366-373 /
 * 416-423) advanced phase by phase together, so that the same commitment of all proofs shares one MSM pass, the same
 * transform one launch per NTT pass, and every proof's lookups / grand products / openings one set of launches.  Proof j
 * gets advice[j * n_advice .. (j + 1) * n_advice) and rng_seeds[32 j .. 32 j + 32) and is byte-identical to zk_prove with
 * the same key, advice and seed; the proofs have one length (*proof_len) and are written proof_stride bytes apart
 * (proofs_out == NULL: only *proof_len).  The first call with a larger batch allocates the further workspaces (kept with
 * the key: ~1.4 GiB each at k = 19).  ZK_EWITNESS means SOME proof's witness is off the table: the batch fails as a whole
 * (zk_prove tells which).  ZK_EINVAL: batch == 0, batch > ZK_PROVE_BATCH_MAX, batch x (#chunks + #lookups) > 256. */
#define ZK_PROVE_BATCH_MAX 64
int zk_prove_batch(zk_ctx* ctx, zk_pk pk, size_t batch, const zk_poly* advice /* batch x n_advice, proof-major */, size_t n_advice,
                   const uint8_t* rng_seeds /* batch x 32 */, int transcript, int scheme, uint8_t* proofs_out, size_t proof_stride,
                   size_t* proof_len);
/* upload canonical (non-Montgomery) integers and convert on the device */
int zk_poly_upload_canonical(zk_ctx* ctx, zk_poly p, const uint64_t* host_canonical, size_t n);

/* ---- timing of the last call of each kind, measured with HIP events on the
 *      context stream (ms); used by bench.py for the roofline figures ---------- */
#define ZK_T_MSM 0
#define ZK_T_NTT 1
#define ZK_T_QUOTIENT 2
#define ZK_T_EVAL 3
#define ZK_T_MSM_ACCUM 4 /* the bucket-accumulation kernel of the last MSM alone */
#define ZK_T_MSM_COLUMNS 5 /* count only: scalar vectors (commitments) the accumulate launches served — a launch
                              serves several columns when commitments are batched */
#define ZK_T_MSM_TAIL_MAIN 6 /* count only: MSM passes whose reduction tail ran on the context's main stream (ZK_OPT_MSM_TAIL_STREAM) */
#define ZK_T_MSM_TAIL 7 /* the reduction tail (T1 .. T3) of the last MSM pass on the wide path, on whichever stream it ran */
#define ZK_T_COUNT 8
int zk_last_kernel_ms(zk_ctx* ctx, int which, float* out_ms);
/* accumulated HIP-event time and launch count since the last reset (ZK_T_MSM, ZK_T_MSM_ACCUM) */
/* ZK_OPT_STREAM_AUDIT: counts[0] = ordering checks made since the option was switched on, counts[1] = violations; msg (may be NULL)
 * receives the description of the first violation (empty if none) */
int zk_audit_report(zk_ctx* ctx, uint64_t counts[2], char* msg, size_t cap);
int zk_timer_reset(zk_ctx* ctx);
/* shader-clock probe: one wave spins for `millis` (1..2000) on a chain of dependent multiply-adds; out[0] = ticks of the shader-clock
 * counter (s_memtime), out[1] = ticks of the constant 100 MHz counter (s_memrealtime) over the same interval, out[2] = multiply-adds
 * issued, out[3] = scratch.  sclk = out[0] / out[1] x 100 MHz.  Called on a context of its own while other contexts prove, it
 * reads the clock the chip sustains UNDER that load (bench.py: roofline.valu_issue) */
int zk_clock_probe(zk_ctx* ctx, uint32_t millis, uint64_t out[4]);
int zk_timer_stats(zk_ctx* ctx, int which, double* total_ms, uint64_t* count);

#ifdef __cplusplus
}
#endif
#endif /* ZKMI355_H */
