/*
 * oracle.c — CPU restatement (plain C, pthreads) of the reference's hot-path
 * operators.  TEST INFRASTRUCTURE ONLY: linked by tests/, smoke() and the
 * cpu_baseline leg of bench.py; never by the product library.
 *
 * The operators live in third-party crates that are not under /root/reference
 * (halo2_proofs PSE fork `arithmetic.rs`, halo2curves `bn256/{fr,fq,curve}.rs`;
 * unpinned git branches — reference halo2-circuits/Cargo.toml:12-15).  They
 * are restated from the published algorithms and reached from the reference at
 * create_proof's call sites, halo2-circuits/src/ecc/ecdsa_p256.rs:366-373,
 * 416-423, 555-562.  Pinning: tests/test_oracle_kat.py (K1, K2, K4 of
 * SURVEY.md §8c) and the golden-proof verifier test (K5).
 *
 * Memory images match the Rust types: Fr/Fq = 4 x u64 little-endian limbs in
 * Montgomery form (R = 2^256); G1Affine = x || y (64 B), identity = (0,0);
 * G1 (Jacobian) = x || y || z (96 B), identity z = 0.
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fe;

typedef struct {
    fe p;          /* modulus */
    uint64_t inv;  /* -p^{-1} mod 2^64 */
    fe r2;         /* R^2 mod p */
    fe one;        /* R mod p */
} field_t;

static const field_t FQ = {
    {{0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL}},
    0x87d20782e4866389ULL,
    {{0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL}},
    {{0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL}},
};
static const field_t FR = {
    {{0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL}},
    0xc2e1f593efffffffULL,
    {{0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL}},
    {{0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL}},
};

/* ---------------------------------------------------------------- field -- */

static inline int fe_is_zero(const fe *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
static inline int fe_eq(const fe *a, const fe *b) { return memcmp(a, b, sizeof(fe)) == 0; }

static inline int fe_geq(const fe *a, const fe *b) {
    for (int i = 3; i >= 0; i--) {
        if (a->l[i] > b->l[i]) return 1;
        if (a->l[i] < b->l[i]) return 0;
    }
    return 1;
}

static inline uint64_t sub_limbs(fe *r, const fe *a, const fe *b) {
    u128 borrow = 0;
    for (int i = 0; i < 4; i++) {
        u128 t = (u128)a->l[i] - b->l[i] - (uint64_t)borrow;
        r->l[i] = (uint64_t)t;
        borrow = (t >> 64) & 1;
    }
    return (uint64_t)borrow;
}

static inline void fe_add(fe *r, const fe *a, const fe *b, const field_t *F) {
    u128 c = 0;
    fe t;
    for (int i = 0; i < 4; i++) {
        c += (u128)a->l[i] + b->l[i];
        t.l[i] = (uint64_t)c;
        c >>= 64;
    }
    /* p < 2^254 so no carry out of 256 bits */
    if (fe_geq(&t, &F->p)) sub_limbs(&t, &t, &F->p);
    *r = t;
}

static inline void fe_sub(fe *r, const fe *a, const fe *b, const field_t *F) {
    fe t;
    if (sub_limbs(&t, a, b)) {
        u128 c = 0;
        for (int i = 0; i < 4; i++) {
            c += (u128)t.l[i] + F->p.l[i];
            t.l[i] = (uint64_t)c;
            c >>= 64;
        }
    }
    *r = t;
}

static inline void fe_neg(fe *r, const fe *a, const field_t *F) {
    if (fe_is_zero(a)) { *r = *a; return; }
    sub_limbs(r, &F->p, a);
}

static inline void fe_dbl(fe *r, const fe *a, const field_t *F) { fe_add(r, a, a, F); }

/* CIOS Montgomery multiplication, 4 x 64-bit limbs (halo2curves field_arithmetic!). */
static inline void fe_mul(fe *r, const fe *a, const fe *b, const field_t *F) {
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)a->l[j] * b->l[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (uint64_t)c;
        t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * F->inv;
        c = (u128)m * F->p.l[0] + t[0];
        c >>= 64;
        for (int j = 1; j < 4; j++) {
            c += (u128)m * F->p.l[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (uint64_t)c;
        t[4] = t[5] + (uint64_t)(c >> 64);
    }
    fe o = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || fe_geq(&o, &F->p)) sub_limbs(&o, &o, &F->p);
    *r = o;
}

static inline void fe_sqr(fe *r, const fe *a, const field_t *F) { fe_mul(r, a, a, F); }

static void fe_pow(fe *r, const fe *a, const fe *e, const field_t *F) {
    fe acc = F->one;
    for (int i = 255; i >= 0; i--) {
        fe_sqr(&acc, &acc, F);
        if ((e->l[i / 64] >> (i % 64)) & 1) fe_mul(&acc, &acc, a, F);
    }
    *r = acc;
}

static void fe_inv(fe *r, const fe *a, const field_t *F) {
    fe e = F->p;
    fe two = {{2, 0, 0, 0}};
    sub_limbs(&e, &e, &two);
    fe_pow(r, a, &e, F);
}

static inline void fe_from_mont(fe *r, const fe *a, const field_t *F) {
    fe one = {{1, 0, 0, 0}};
    fe_mul(r, a, &one, F);
}

static inline void fe_to_mont(fe *r, const fe *a, const field_t *F) { fe_mul(r, a, &F->r2, F); }

/* ------------------------------------------------------------------- G1 -- */

typedef struct { fe x, y; } g1a;      /* affine, identity = (0,0) */
typedef struct { fe x, y, z; } g1j;   /* Jacobian, identity z = 0 */

static inline int g1a_is_id(const g1a *p) { return fe_is_zero(&p->x) && fe_is_zero(&p->y); }
static inline void g1j_set_id(g1j *p) { memset(p, 0, sizeof(*p)); p->x = FQ.one; p->y = FQ.one; }

static void g1j_dbl(g1j *r, const g1j *p) {
    if (fe_is_zero(&p->z)) { *r = *p; return; }
    const field_t *F = &FQ;
    fe a, b, c, d, e, f, t, x3, y3, z3;
    fe_sqr(&a, &p->x, F);
    fe_sqr(&b, &p->y, F);
    fe_sqr(&c, &b, F);
    fe_add(&t, &p->x, &b, F); fe_sqr(&t, &t, F); fe_sub(&t, &t, &a, F); fe_sub(&t, &t, &c, F);
    fe_dbl(&d, &t, F);
    fe_dbl(&e, &a, F); fe_add(&e, &e, &a, F);
    fe_sqr(&f, &e, F);
    fe_dbl(&t, &d, F); fe_sub(&x3, &f, &t, F);
    fe_mul(&z3, &p->y, &p->z, F); fe_dbl(&z3, &z3, F);
    fe_sub(&t, &d, &x3, F); fe_mul(&y3, &e, &t, F);
    fe_dbl(&c, &c, F); fe_dbl(&c, &c, F); fe_dbl(&c, &c, F);
    fe_sub(&y3, &y3, &c, F);
    r->x = x3; r->y = y3; r->z = z3;
}

/* r = p + q (q affine) */
static void g1j_add_mixed(g1j *r, const g1j *p, const g1a *q) {
    const field_t *F = &FQ;
    if (g1a_is_id(q)) { *r = *p; return; }
    if (fe_is_zero(&p->z)) { r->x = q->x; r->y = q->y; r->z = F->one; return; }
    fe z1z1, u2, s2, h, hh, i, j, rr, v, t, x3, y3, z3;
    fe_sqr(&z1z1, &p->z, F);
    fe_mul(&u2, &q->x, &z1z1, F);
    fe_mul(&s2, &q->y, &p->z, F); fe_mul(&s2, &s2, &z1z1, F);
    if (fe_eq(&u2, &p->x)) {
        if (fe_eq(&s2, &p->y)) { g1j_dbl(r, p); return; }
        g1j_set_id(r); return;
    }
    fe_sub(&h, &u2, &p->x, F);
    fe_sqr(&hh, &h, F);
    fe_dbl(&i, &hh, F); fe_dbl(&i, &i, F);
    fe_mul(&j, &h, &i, F);
    fe_sub(&rr, &s2, &p->y, F); fe_dbl(&rr, &rr, F);
    fe_mul(&v, &p->x, &i, F);
    fe_sqr(&x3, &rr, F); fe_sub(&x3, &x3, &j, F); fe_dbl(&t, &v, F); fe_sub(&x3, &x3, &t, F);
    fe_sub(&t, &v, &x3, F); fe_mul(&y3, &rr, &t, F);
    fe_mul(&t, &p->y, &j, F); fe_dbl(&t, &t, F); fe_sub(&y3, &y3, &t, F);
    fe_add(&z3, &p->z, &h, F); fe_sqr(&z3, &z3, F); fe_sub(&z3, &z3, &z1z1, F); fe_sub(&z3, &z3, &hh, F);
    r->x = x3; r->y = y3; r->z = z3;
}

static void g1j_add(g1j *r, const g1j *p, const g1j *q) {
    const field_t *F = &FQ;
    if (fe_is_zero(&p->z)) { *r = *q; return; }
    if (fe_is_zero(&q->z)) { *r = *p; return; }
    fe z1z1, z2z2, u1, u2, s1, s2, h, i, j, rr, v, t, x3, y3, z3;
    fe_sqr(&z1z1, &p->z, F);
    fe_sqr(&z2z2, &q->z, F);
    fe_mul(&u1, &p->x, &z2z2, F);
    fe_mul(&u2, &q->x, &z1z1, F);
    fe_mul(&s1, &p->y, &q->z, F); fe_mul(&s1, &s1, &z2z2, F);
    fe_mul(&s2, &q->y, &p->z, F); fe_mul(&s2, &s2, &z1z1, F);
    if (fe_eq(&u1, &u2)) {
        if (fe_eq(&s1, &s2)) { g1j_dbl(r, p); return; }
        g1j_set_id(r); return;
    }
    fe_sub(&h, &u2, &u1, F);
    fe_dbl(&i, &h, F); fe_sqr(&i, &i, F);
    fe_mul(&j, &h, &i, F);
    fe_sub(&rr, &s2, &s1, F); fe_dbl(&rr, &rr, F);
    fe_mul(&v, &u1, &i, F);
    fe_sqr(&x3, &rr, F); fe_sub(&x3, &x3, &j, F); fe_dbl(&t, &v, F); fe_sub(&x3, &x3, &t, F);
    fe_sub(&t, &v, &x3, F); fe_mul(&y3, &rr, &t, F);
    fe_mul(&t, &s1, &j, F); fe_dbl(&t, &t, F); fe_sub(&y3, &y3, &t, F);
    fe_add(&z3, &p->z, &q->z, F); fe_sqr(&z3, &z3, F); fe_sub(&z3, &z3, &z1z1, F); fe_sub(&z3, &z3, &z2z2, F);
    fe_mul(&z3, &z3, &h, F);
    r->x = x3; r->y = y3; r->z = z3;
}

static void g1j_to_affine(g1a *r, const g1j *p) {
    const field_t *F = &FQ;
    if (fe_is_zero(&p->z)) { memset(r, 0, sizeof(*r)); return; }
    fe zi, zi2, zi3;
    fe_inv(&zi, &p->z, F);
    fe_sqr(&zi2, &zi, F);
    fe_mul(&zi3, &zi2, &zi, F);
    fe_mul(&r->x, &p->x, &zi2, F);
    fe_mul(&r->y, &p->y, &zi3, F);
}

/* ------------------------------------------------------------- threading -- */

typedef void (*range_fn)(void *ctx, size_t tid, size_t lo, size_t hi);
typedef struct { range_fn fn; void *ctx; size_t tid, lo, hi; } job_t;
static void *job_tramp(void *p) { job_t *j = p; j->fn(j->ctx, j->tid, j->lo, j->hi); return NULL; }

static void parallel_for(size_t n, int nthreads, range_fn fn, void *ctx) {
    if (nthreads < 1) nthreads = 1;
    if ((size_t)nthreads > n) nthreads = n ? (int)n : 1;
    if (nthreads == 1) { fn(ctx, 0, 0, n); return; }
    pthread_t *th = malloc(sizeof(pthread_t) * nthreads);
    job_t *jobs = malloc(sizeof(job_t) * nthreads);
    size_t chunk = (n + nthreads - 1) / nthreads;
    int started = 0;
    for (int t = 0; t < nthreads; t++) {
        size_t lo = (size_t)t * chunk, hi = lo + chunk;
        if (lo >= n) break;
        if (hi > n) hi = n;
        jobs[t] = (job_t){fn, ctx, (size_t)t, lo, hi};
        pthread_create(&th[t], NULL, job_tramp, &jobs[t]);
        started++;
    }
    for (int t = 0; t < started; t++) pthread_join(th[t], NULL);
    free(th); free(jobs);
}

/* ------------------------------------------------------------------ MSM -- */
/* Restates halo2_proofs arithmetic.rs `multiexp_serial` / `best_multiexp`:
 * one chunk per thread, each a serial Pippenger with window c = ceil(ln n),
 * (256/c)+1 segments, 2^c - 1 buckets, running-sum reduction; chunk results
 * are added in order. */

static inline unsigned get_at(unsigned seg, unsigned c, const uint8_t *bytes) {
    unsigned skip_bits = seg * c, skip_bytes = skip_bits / 8;
    if (skip_bytes >= 32) return 0;
    uint64_t v = 0;
    for (unsigned i = 0; i < 8 && skip_bytes + i < 32; i++) v |= (uint64_t)bytes[skip_bytes + i] << (8 * i);
    v >>= skip_bits - skip_bytes * 8;
    return (unsigned)(v % (1ull << c));
}

static void multiexp_serial(const fe *coeffs_mont, const g1a *bases, size_t n, g1j *acc) {
    uint8_t *repr = malloc(n * 32);
    for (size_t i = 0; i < n; i++) {
        fe t; fe_from_mont(&t, &coeffs_mont[i], &FR);
        memcpy(repr + 32 * i, t.l, 32);
    }
    unsigned c = n < 4 ? 1 : n < 32 ? 3 : (unsigned)ceil(log((double)n));
    unsigned segments = 256 / c + 1;
    size_t nb = ((size_t)1 << c) - 1;
    g1j *buckets = malloc(sizeof(g1j) * nb);
    for (int seg = (int)segments - 1; seg >= 0; seg--) {
        for (unsigned i = 0; i < c; i++) g1j_dbl(acc, acc);
        for (size_t b = 0; b < nb; b++) g1j_set_id(&buckets[b]);
        for (size_t i = 0; i < n; i++) {
            unsigned d = get_at((unsigned)seg, c, repr + 32 * i);
            if (d) g1j_add_mixed(&buckets[d - 1], &buckets[d - 1], &bases[i]);
        }
        g1j running; g1j_set_id(&running);
        for (size_t b = nb; b-- > 0;) {
            g1j_add(&running, &running, &buckets[b]);
            g1j_add(acc, acc, &running);
        }
    }
    free(buckets); free(repr);
}

typedef struct { const fe *s; const g1a *b; g1j *res; } msm_ctx;
static void msm_job(void *p, size_t tid, size_t lo, size_t hi) {
    msm_ctx *c = p;
    g1j_set_id(&c->res[tid]);
    multiexp_serial(c->s + lo, c->b + lo, hi - lo, &c->res[tid]);
}

int orc_msm_bn254(const uint64_t *scalars_mont, const uint64_t *bases_affine_mont, size_t n,
                  uint64_t out_jacobian_mont[12], int nthreads) {
    g1j acc; g1j_set_id(&acc);
    if (nthreads < 1) nthreads = 1;
    if (n > (size_t)nthreads && nthreads > 1) {
        g1j *res = malloc(sizeof(g1j) * nthreads);
        for (int t = 0; t < nthreads; t++) g1j_set_id(&res[t]);
        msm_ctx ctx = {(const fe *)scalars_mont, (const g1a *)bases_affine_mont, res};
        parallel_for(n, nthreads, msm_job, &ctx);
        for (int t = 0; t < nthreads; t++) g1j_add(&acc, &acc, &res[t]);
        free(res);
    } else {
        multiexp_serial((const fe *)scalars_mont, (const g1a *)bases_affine_mont, n, &acc);
    }
    memcpy(out_jacobian_mont, &acc, 96);
    return 0;
}

/* ------------------------------------------------------------------ NTT -- */
/* Restates halo2_proofs arithmetic.rs `best_fft`: bit-reverse swap, then
 * radix-2 DIT butterflies with precomputed twiddles w^i; natural order in/out.*/

static inline uint32_t bitrev(uint32_t x, unsigned bits) {
    uint32_t r = 0;
    for (unsigned i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; }
    return r;
}

typedef struct { fe *a; const fe *tw; size_t n; size_t half; } ntt_ctx;
static void ntt_stage_job(void *p, size_t tid, size_t lo, size_t hi) {
    (void)tid;
    ntt_ctx *c = p;
    size_t half = c->half, step = c->n / (2 * half);
    for (size_t bf = lo; bf < hi; bf++) {
        size_t blk = bf / half, j = bf % half;
        fe *u = &c->a[blk * 2 * half + j], *v = u + half;
        fe t; fe_mul(&t, v, &c->tw[j * step], &FR);
        fe_sub(v, u, &t, &FR);
        fe_add(u, u, &t, &FR);
    }
}

int orc_ntt_bn254_fr(uint64_t *a_mont, const uint64_t omega_mont[4], uint32_t log_n, int nthreads) {
    size_t n = (size_t)1 << log_n;
    fe *a = (fe *)a_mont;
    for (size_t i = 0; i < n; i++) {
        size_t r = bitrev((uint32_t)i, log_n);
        if (i < r) { fe t = a[i]; a[i] = a[r]; a[r] = t; }
    }
    fe *tw = malloc(sizeof(fe) * (n / 2 ? n / 2 : 1));
    fe w; memcpy(&w, omega_mont, 32);
    tw[0] = FR.one;
    for (size_t i = 1; i < n / 2; i++) fe_mul(&tw[i], &tw[i - 1], &w, &FR);
    for (size_t half = 1; half < n; half <<= 1) {
        ntt_ctx ctx = {a, tw, n, half};
        parallel_for(n / 2, nthreads, ntt_stage_job, &ctx);
    }
    free(tw);
    return 0;
}

/* --------------------------------------------------- small exported ops -- */

static const field_t *pick(int which) { return which ? &FQ : &FR; }  /* 0 = Fr, 1 = Fq */

void orc_fe_mul(int which, const uint64_t *a, const uint64_t *b, uint64_t *r) { fe_mul((fe *)r, (const fe *)a, (const fe *)b, pick(which)); }
void orc_fe_add(int which, const uint64_t *a, const uint64_t *b, uint64_t *r) { fe_add((fe *)r, (const fe *)a, (const fe *)b, pick(which)); }
void orc_fe_sub(int which, const uint64_t *a, const uint64_t *b, uint64_t *r) { fe_sub((fe *)r, (const fe *)a, (const fe *)b, pick(which)); }
void orc_fe_inv(int which, const uint64_t *a, uint64_t *r) { fe_inv((fe *)r, (const fe *)a, pick(which)); }
void orc_fe_to_mont(int which, const uint64_t *a, uint64_t *r, size_t n) {
    for (size_t i = 0; i < n; i++) fe_to_mont((fe *)(r + 4 * i), (const fe *)(a + 4 * i), pick(which));
}
void orc_fe_from_mont(int which, const uint64_t *a, uint64_t *r, size_t n) {
    for (size_t i = 0; i < n; i++) fe_from_mont((fe *)(r + 4 * i), (const fe *)(a + 4 * i), pick(which));
}
void orc_g1_to_affine(const uint64_t *jac, uint64_t *aff) { g1j_to_affine((g1a *)aff, (const g1j *)jac); }
void orc_g1_add_mixed(const uint64_t *jac, const uint64_t *aff, uint64_t *out) { g1j_add_mixed((g1j *)out, (const g1j *)jac, (const g1a *)aff); }
void orc_g1_add(const uint64_t *p, const uint64_t *q, uint64_t *out) { g1j_add((g1j *)out, (const g1j *)p, (const g1j *)q); }
void orc_g1_dbl(const uint64_t *p, uint64_t *out) { g1j_dbl((g1j *)out, (const g1j *)p); }

/* Fixed-base multiples of the G1 generator: out[i] = [s_i] G, affine Montgomery.
 * Used to build SRS fixtures (g[i] = [tau^i]G, g_lagrange[i] = [L_i(tau)]G —
 * ParamsKZG::setup with the secret known).  Window 8. */
typedef struct { const fe *s; g1a *out; const g1a *table; } fb_ctx;
static void fb_job(void *p, size_t tid, size_t lo, size_t hi) {
    (void)tid;
    fb_ctx *c = p;
    size_t cnt = hi - lo;
    g1j *acc = malloc(sizeof(g1j) * cnt);
    for (size_t i = 0; i < cnt; i++) {
        fe k; fe_from_mont(&k, &c->s[lo + i], &FR);
        g1j_set_id(&acc[i]);
        const uint8_t *kb = (const uint8_t *)k.l;
        for (int w = 0; w < 32; w++)
            if (kb[w]) g1j_add_mixed(&acc[i], &acc[i], &c->table[w * 256 + kb[w]]);
    }
    /* batch normalisation (Montgomery's trick) */
    fe *pref = malloc(sizeof(fe) * (cnt + 1));
    pref[0] = FQ.one;
    for (size_t i = 0; i < cnt; i++) {
        if (fe_is_zero(&acc[i].z)) pref[i + 1] = pref[i];
        else fe_mul(&pref[i + 1], &pref[i], &acc[i].z, &FQ);
    }
    fe inv; fe_inv(&inv, &pref[cnt], &FQ);
    for (size_t i = cnt; i-- > 0;) {
        g1a *o = &c->out[lo + i];
        if (fe_is_zero(&acc[i].z)) { memset(o, 0, sizeof(*o)); continue; }
        fe zi, zi2, zi3;
        fe_mul(&zi, &inv, &pref[i], &FQ);
        fe_mul(&inv, &inv, &acc[i].z, &FQ);
        fe_sqr(&zi2, &zi, &FQ); fe_mul(&zi3, &zi2, &zi, &FQ);
        fe_mul(&o->x, &acc[i].x, &zi2, &FQ);
        fe_mul(&o->y, &acc[i].y, &zi3, &FQ);
    }
    free(pref); free(acc);
}

int orc_fixed_base_g1(const uint64_t *scalars_mont, size_t n, uint64_t *out_affine_mont, int nthreads) {
    g1a *table = malloc(sizeof(g1a) * 32 * 256);
    g1j base; base.x = FQ.one; fe two = {{2, 0, 0, 0}}; fe_to_mont(&base.y, &two, &FQ); base.z = FQ.one;
    for (int w = 0; w < 32; w++) {
        g1j cur; g1j_set_id(&cur);
        memset(&table[w * 256], 0, sizeof(g1a));
        for (int d = 1; d < 256; d++) {
            g1j_add(&cur, &cur, &base);
            g1j_to_affine(&table[w * 256 + d], &cur);
        }
        g1j_add(&base, &cur, &base); /* 256 * base */
    }
    fb_ctx ctx = {(const fe *)scalars_mont, (g1a *)out_affine_mont, table};
    parallel_for(n, nthreads, fb_job, &ctx);
    free(table);
    return 0;
}

/* powers: out[i] = base^i * first (Montgomery in/out) */
void orc_fr_powers(const uint64_t *base_mont, const uint64_t *first_mont, uint64_t *out, size_t n) {
    fe cur; memcpy(&cur, first_mont, 32);
    fe b; memcpy(&b, base_mont, 32);
    for (size_t i = 0; i < n; i++) { memcpy(out + 4 * i, &cur, 32); fe_mul(&cur, &cur, &b, &FR); }
}

/* ===================================================== prover vector ops ==
 * Element-wise / scan operators of halo2_proofs' create_proof between its MSM and FFT calls
 * (plonk/prover.rs, plonk/permutation/prover.rs, plonk/lookup/prover.rs, plonk/evaluation.rs,
 * arithmetic.rs `eval_polynomial` / `kate_division`; reached from the reference at
 * halo2-circuits/src/ecc/ecdsa_p256.rs:366-373, 416-423), restated for the oracle's full-size
 * CPU prover (oracle/zkoracle/fastprover.py), which is (a) the generator of the BASELINE-size golden
 * proofs under tests/golden/ and (b) bench.py's whole-proof cpu_baseline.  halo2 parallelises these
 * loops with rayon over rows; here: pthreads over rows.  All vectors are Fr, Montgomery form. */

typedef struct { fe *out; const fe *a, *b; fe ca, cb, k; int has_b; } lin_ctx;
static void lin_job(void *p, size_t tid, size_t lo, size_t hi) {
    (void)tid;
    lin_ctx *c = p;
    for (size_t i = lo; i < hi; i++) {
        fe t; fe_mul(&t, &c->a[i], &c->ca, &FR);
        if (c->has_b) { fe u; fe_mul(&u, &c->b[i], &c->cb, &FR); fe_add(&t, &t, &u, &FR); }
        fe_add(&c->out[i], &t, &c->k, &FR);
    }
}
/* out[i] = ca a[i] + cb b[i] + k  (b may be NULL) */
void orc_vec_lin(uint64_t *out, const uint64_t *a, const uint64_t *ca, const uint64_t *b, const uint64_t *cb, const uint64_t *k,
                 size_t n, int nthreads) {
    lin_ctx c = {(fe *)out, (const fe *)a, (const fe *)b, *(const fe *)ca, b ? *(const fe *)cb : FR.one, *(const fe *)k, b != NULL};
    parallel_for(n, nthreads, lin_job, &c);
}

typedef struct { fe *out; const fe *a, *b; } mul_ctx;
static void mul_job(void *p, size_t tid, size_t lo, size_t hi) {
    (void)tid;
    mul_ctx *c = p;
    for (size_t i = lo; i < hi; i++) fe_mul(&c->out[i], &c->a[i], &c->b[i], &FR);
}
void orc_vec_mul(uint64_t *out, const uint64_t *a, const uint64_t *b, size_t n, int nthreads) {
    mul_ctx c = {(fe *)out, (const fe *)a, (const fe *)b};
    parallel_for(n, nthreads, mul_job, &c);
}

typedef struct { fe *a; fe s[3]; } p3_ctx;
static void p3_job(void *p, size_t tid, size_t lo, size_t hi) {
    (void)tid;
    p3_ctx *c = p;
    for (size_t i = lo; i < hi; i++) fe_mul(&c->a[i], &c->a[i], &c->s[i % 3], &FR);
}
/* a[i] *= s[i mod 3]: EvaluationDomain::distribute_powers_zeta (coset scaling by the cube root of unity), times a constant */
void orc_vec_scale_period3(uint64_t *a, const uint64_t *s3, size_t n, int nthreads) {
    p3_ctx c; c.a = (fe *)a; memcpy(c.s, s3, 96);
    parallel_for(n, nthreads, p3_job, &c);
}

/* halo2 `batch_invert` semantics over a whole vector: out[i] = a[i]^-1, 0 -> 0 (Montgomery's trick per thread chunk) */
typedef struct { fe *out; const fe *a; } binv_ctx;
static void binv_job(void *p, size_t tid, size_t lo, size_t hi) {
    (void)tid;
    binv_ctx *c = p;
    size_t cnt = hi - lo;
    fe *pref = malloc(sizeof(fe) * (cnt + 1));
    pref[0] = FR.one;
    for (size_t i = 0; i < cnt; i++) {
        if (fe_is_zero(&c->a[lo + i])) pref[i + 1] = pref[i];
        else fe_mul(&pref[i + 1], &pref[i], &c->a[lo + i], &FR);
    }
    fe inv; fe_inv(&inv, &pref[cnt], &FR);
    for (size_t i = cnt; i-- > 0;) {
        if (fe_is_zero(&c->a[lo + i])) { memset(&c->out[lo + i], 0, sizeof(fe)); continue; }
        fe ai = c->a[lo + i];  /* out may alias a */
        fe_mul(&c->out[lo + i], &inv, &pref[i], &FR);
        fe_mul(&inv, &inv, &ai, &FR);
    }
    free(pref);
}
void orc_vec_batch_inv(uint64_t *out, const uint64_t *a, size_t n, int nthreads) {
    binv_ctx c = {(fe *)out, (const fe *)a};
    parallel_for(n, nthreads, binv_job, &c);
}

/* z[0] = init, z[i + 1] = z[i] * f[i] for i + 1 < n  (the grand-product running product; serial, as in halo2) */
void orc_running_product(uint64_t *z, const uint64_t *f, const uint64_t *init, size_t n) {
    fe *zz = (fe *)z; const fe *ff = (const fe *)f;
    memcpy(&zz[0], init, 32);
    for (size_t i = 0; i + 1 < n; i++) fe_mul(&zz[i + 1], &zz[i], &ff[i], &FR);
}

/* sum_i a[i] b[i] */
typedef struct { const fe *a, *b; fe *part; } dot_ctx;
static void dot_job(void *p, size_t tid, size_t lo, size_t hi) {
    dot_ctx *c = p;
    fe acc; memset(&acc, 0, sizeof(acc));
    for (size_t i = lo; i < hi; i++) { fe t; fe_mul(&t, &c->a[i], &c->b[i], &FR); fe_add(&acc, &acc, &t, &FR); }
    c->part[tid] = acc;
}
void orc_vec_dot(const uint64_t *a, const uint64_t *b, size_t n, uint64_t *out, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    fe *part = calloc((size_t)nthreads, sizeof(fe));
    dot_ctx c = {(const fe *)a, (const fe *)b, part};
    parallel_for(n, nthreads, dot_job, &c);
    fe acc; memset(&acc, 0, sizeof(acc));
    for (int t = 0; t < nthreads; t++) fe_add(&acc, &acc, &part[t], &FR);
    memcpy(out, &acc, 32);
    free(part);
}

/* arithmetic.rs eval_polynomial: Horner per thread chunk, chunks combined with x^chunk */
typedef struct { const fe *c; fe x; fe *part; size_t *len; } ev_ctx;
static void ev_job(void *p, size_t tid, size_t lo, size_t hi) {
    ev_ctx *c = p;
    fe acc; memset(&acc, 0, sizeof(acc));
    for (size_t i = hi; i-- > lo;) { fe_mul(&acc, &acc, &c->x, &FR); fe_add(&acc, &acc, &c->c[i], &FR); }
    c->part[tid] = acc;
    c->len[tid] = hi - lo;
}
void orc_eval_poly(const uint64_t *coeffs, size_t n, const uint64_t *x, uint64_t *out, int nthreads) {
    if (nthreads < 1) nthreads = 1;
    fe *part = calloc((size_t)nthreads, sizeof(fe));
    size_t *len = calloc((size_t)nthreads, sizeof(size_t));
    ev_ctx c; c.c = (const fe *)coeffs; memcpy(&c.x, x, 32); c.part = part; c.len = len;
    parallel_for(n, nthreads, ev_job, &c);
    /* value = sum_t part[t] * x^(offset_t): combine from the top chunk down */
    fe acc; memset(&acc, 0, sizeof(acc));
    for (int t = nthreads - 1; t >= 0; t--) {
        if (!len[t]) continue;
        /* acc = acc * x^len[t] + part[t] */
        fe pw = FR.one, b = c.x;
        for (size_t e = len[t]; e; e >>= 1) { if (e & 1) fe_mul(&pw, &pw, &b, &FR); fe_mul(&b, &b, &b, &FR); }
        fe_mul(&acc, &acc, &pw, &FR);
        fe_add(&acc, &acc, &part[t], &FR);
    }
    memcpy(out, &acc, 32);
    free(part); free(len);
}

/* arithmetic.rs kate_division: (p(X) - p(z)) / (X - z); n coefficients in, n - 1 out (q[n - 1] is set to 0) */
void orc_kate_division(const uint64_t *p, size_t n, const uint64_t *z, uint64_t *q) {
    const fe *pp = (const fe *)p; fe *qq = (fe *)q;
    fe zz; memcpy(&zz, z, 32);
    fe carry; memset(&carry, 0, sizeof(carry));
    memset(&qq[n - 1], 0, sizeof(fe));
    for (size_t i = n - 1; i >= 1; i--) {
        fe t; fe_mul(&t, &carry, &zz, &FR);
        fe_add(&carry, &pp[i], &t, &FR);
        qq[i - 1] = carry;
    }
}

/* plonk/evaluation.rs Evaluator::evaluate_h for the halo2-lib column shape + divide_by_vanishing_poly, one row of
 * the extended coset per iteration.  Expressions and y-Horner order as the reference's generated verifier checks
 * them (proving-server/P256Verifier.yul:406-552).  Pointers are extended-coset vectors of 2^log_ext rows. */
typedef struct {
    uint32_t log_ext, n_gate, n_chunks, chunk_len, n_perm, n_lookups, single, fx_table, fx_qlookup;
    int32_t last_rot;
    const fe *const *adv; const fe *const *fix; const int32_t *fx_sel;
    const fe *const *sigma; const fe *const *perm_val; const fe *const *z;
    const fe *const *lk_z; const fe *const *lk_a; const fe *const *lk_s; const fe *const *lk_in;
    const fe *l0, *l_last, *l_blind, *xs;
    fe beta, gamma, y;
    const fe *delta_pow;  /* delta^p */
    fe t_inv[4];
    fe *out;
} quot_ctx;

static void quot_job(void *p, size_t tid, size_t lo, size_t hi) {
    (void)tid;
    const quot_ctx *a = p;
    const size_t N = (size_t)1 << a->log_ext, mask = N - 1;
    for (size_t i = lo; i < hi; i++) {
#define ROT(r) ((i + (size_t)((int64_t)(r) * 4)) & mask)
        fe acc; memset(&acc, 0, sizeof(acc));
#define PUSH(e) do { fe_mul(&acc, &acc, &a->y, &FR); fe_add(&acc, &acc, (e), &FR); } while (0)
        fe t, u, v;
        for (uint32_t j = 0; j < a->n_gate; j++) {
            /* fx_sel[j] = fixed column | form << 24: the gate's selector after halo2's compress_selectors
             * (zkoracle/plonk.py Shape.gate_sel): form 0: q, 1: q (2 - q), 2: q (1 - q) */
            const fe *c = a->adv[j];
            const fe *q = &a->fix[a->fx_sel[j] & 0xffffff][i];
            const int form = a->fx_sel[j] >> 24;
            fe_mul(&t, &c[ROT(1)], &c[ROT(2)], &FR);
            fe_add(&t, &t, &c[i], &FR);
            fe_sub(&t, &t, &c[ROT(3)], &FR);
            fe_mul(&t, &t, q, &FR);
            if (form) {
                fe_sub(&u, &FR.one, q, &FR);
                if (form == 1) fe_add(&u, &u, &FR.one, &FR);
                fe_mul(&t, &t, &u, &FR);
            }
            PUSH(&t);
        }
        const fe *l0 = &a->l0[i], *ll = &a->l_last[i];
        fe active; fe_sub(&active, &FR.one, ll, &FR); fe_sub(&active, &active, &a->l_blind[i], &FR);
        /* permutation */
        fe_sub(&t, &FR.one, &a->z[0][i], &FR); fe_mul(&t, &t, l0, &FR); PUSH(&t);
        { const fe *zl = &a->z[a->n_chunks - 1][i]; fe_mul(&t, zl, zl, &FR); fe_sub(&t, &t, zl, &FR); fe_mul(&t, &t, ll, &FR); PUSH(&t); }
        for (uint32_t c = 1; c < a->n_chunks; c++) {
            fe_sub(&t, &a->z[c][i], &a->z[c - 1][ROT(a->last_rot)], &FR); fe_mul(&t, &t, l0, &FR); PUSH(&t);
        }
        fe bx; fe_mul(&bx, &a->beta, &a->xs[i], &FR);
        for (uint32_t c = 0; c < a->n_chunks; c++) {
            fe left = a->z[c][ROT(1)], right = a->z[c][i];
            uint32_t lo_p = c * a->chunk_len, hi_p = lo_p + a->chunk_len; if (hi_p > a->n_perm) hi_p = a->n_perm;
            for (uint32_t q = lo_p; q < hi_p; q++) {
                fe vg; fe_add(&vg, &a->perm_val[q][i], &a->gamma, &FR);
                fe_mul(&u, &a->beta, &a->sigma[q][i], &FR); fe_add(&u, &u, &vg, &FR); fe_mul(&left, &left, &u, &FR);
                fe_mul(&v, &bx, &a->delta_pow[q], &FR); fe_add(&v, &v, &vg, &FR); fe_mul(&right, &right, &v, &FR);
            }
            fe_sub(&t, &left, &right, &FR); fe_mul(&t, &t, &active, &FR); PUSH(&t);
        }
        /* lookups */
        for (uint32_t l = 0; l < a->n_lookups; l++) {
            const fe *zc = &a->lk_z[l][i], *zn = &a->lk_z[l][ROT(1)];
            const fe *ap = &a->lk_a[l][i], *apm = &a->lk_a[l][ROT(-1)], *sp = &a->lk_s[l][i];
            fe inp;
            if (a->single) fe_mul(&inp, &a->fix[a->fx_qlookup][i], &a->adv[0][i], &FR); else inp = a->lk_in[l][i];
            const fe *tab = &a->fix[a->fx_table][i];
            fe_sub(&t, &FR.one, zc, &FR); fe_mul(&t, &t, l0, &FR); PUSH(&t);
            fe_mul(&t, zc, zc, &FR); fe_sub(&t, &t, zc, &FR); fe_mul(&t, &t, ll, &FR); PUSH(&t);
            fe left, right;
            fe_add(&u, ap, &a->beta, &FR); fe_mul(&left, zn, &u, &FR); fe_add(&u, sp, &a->gamma, &FR); fe_mul(&left, &left, &u, &FR);
            fe_add(&u, &inp, &a->beta, &FR); fe_mul(&right, zc, &u, &FR); fe_add(&u, tab, &a->gamma, &FR); fe_mul(&right, &right, &u, &FR);
            fe_sub(&t, &left, &right, &FR); fe_mul(&t, &t, &active, &FR); PUSH(&t);
            fe d; fe_sub(&d, ap, sp, &FR);
            fe_mul(&t, &d, l0, &FR); PUSH(&t);
            fe_sub(&u, ap, apm, &FR); fe_mul(&t, &d, &u, &FR); fe_mul(&t, &t, &active, &FR); PUSH(&t);
        }
        fe_mul(&a->out[i], &acc, &a->t_inv[i & 3], &FR);
#undef ROT
#undef PUSH
    }
}

int orc_quotient(uint32_t log_ext, uint32_t n_gate, uint32_t n_chunks, uint32_t chunk_len, uint32_t n_perm, uint32_t n_lookups,
                 uint32_t single, uint32_t fx_table, uint32_t fx_qlookup, int32_t last_rot,
                 const uint64_t *const *adv, const uint64_t *const *fix, const int32_t *fx_sel,
                 const uint64_t *const *sigma, const uint64_t *const *perm_val, const uint64_t *const *z,
                 const uint64_t *const *lk_z, const uint64_t *const *lk_a, const uint64_t *const *lk_s, const uint64_t *const *lk_in,
                 const uint64_t *l0, const uint64_t *l_last, const uint64_t *l_blind, const uint64_t *xs,
                 const uint64_t *beta, const uint64_t *gamma, const uint64_t *y, const uint64_t *delta_pow, const uint64_t *t_inv4,
                 uint64_t *out, int nthreads) {
    quot_ctx c;
    c.log_ext = log_ext; c.n_gate = n_gate; c.n_chunks = n_chunks; c.chunk_len = chunk_len; c.n_perm = n_perm;
    c.n_lookups = n_lookups; c.single = single; c.fx_table = fx_table; c.fx_qlookup = fx_qlookup; c.last_rot = last_rot;
    c.adv = (const fe *const *)adv; c.fix = (const fe *const *)fix; c.fx_sel = fx_sel;
    c.sigma = (const fe *const *)sigma; c.perm_val = (const fe *const *)perm_val; c.z = (const fe *const *)z;
    c.lk_z = (const fe *const *)lk_z; c.lk_a = (const fe *const *)lk_a; c.lk_s = (const fe *const *)lk_s; c.lk_in = (const fe *const *)lk_in;
    c.l0 = (const fe *)l0; c.l_last = (const fe *)l_last; c.l_blind = (const fe *)l_blind; c.xs = (const fe *)xs;
    memcpy(&c.beta, beta, 32); memcpy(&c.gamma, gamma, 32); memcpy(&c.y, y, 32);
    c.delta_pow = (const fe *)delta_pow; memcpy(c.t_inv, t_inv4, 128);
    c.out = (fe *)out;
    parallel_for((size_t)1 << log_ext, nthreads, quot_job, &c);
    return 0;
}

/* rand_chacha ChaCha20Rng keystream (key = seed, nonce / stream 0, 64-bit block counter) -> one Fr per 64-byte
 * block: halo2curves `Fr::random` = from_u512(eight next_u64).  out[i] = block first + i. */
static inline uint32_t rotl32(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
#define ORC_QR(a, b, c, d) a += b; d ^= a; d = rotl32(d, 16); c += d; b ^= c; b = rotl32(b, 12); a += b; d ^= a; d = rotl32(d, 8); c += d; b ^= c; b = rotl32(b, 7);
static void chacha20_block(const uint8_t key[32], uint64_t counter, uint8_t out[64]) {
    uint32_t st[16] = {0x61707865, 0x3320646e, 0x79622d32, 0x6b206574};
    memcpy(st + 4, key, 32);
    st[12] = (uint32_t)counter; st[13] = (uint32_t)(counter >> 32); st[14] = 0; st[15] = 0;
    uint32_t w[16];
    memcpy(w, st, 64);
    for (int i = 0; i < 10; i++) {
        ORC_QR(w[0], w[4], w[8], w[12]) ORC_QR(w[1], w[5], w[9], w[13]) ORC_QR(w[2], w[6], w[10], w[14]) ORC_QR(w[3], w[7], w[11], w[15])
        ORC_QR(w[0], w[5], w[10], w[15]) ORC_QR(w[1], w[6], w[11], w[12]) ORC_QR(w[2], w[7], w[8], w[13]) ORC_QR(w[3], w[4], w[9], w[14])
    }
    for (int i = 0; i < 16; i++) w[i] += st[i];
    memcpy(out, w, 64);
}
typedef struct { const uint8_t *key; uint64_t first; fe *out; fe r3; } cc_ctx;
static void cc_job(void *p, size_t tid, size_t lo, size_t hi) {
    (void)tid;
    cc_ctx *c = p;
    for (size_t i = lo; i < hi; i++) {
        uint8_t b[64];
        chacha20_block(c->key, c->first + i, b);
        fe l, h, t, u;
        memcpy(&l, b, 32); memcpy(&h, b + 32, 32);
        /* (l + h 2^256) R mod r = l R + h R^2: CIOS tolerates one operand < 2^256 when the other is < r */
        fe_mul(&t, &l, &FR.r2, &FR);
        fe_mul(&u, &h, &c->r3, &FR);
        fe_add(&c->out[i], &t, &u, &FR);
    }
}
void orc_chacha20_fr(const uint8_t key[32], uint64_t first_block, size_t count, uint64_t *out_mont, int nthreads) {
    cc_ctx c; c.key = key; c.first = first_block; c.out = (fe *)out_mont;
    fe_mul(&c.r3, &FR.r2, &FR.r2, &FR);
    parallel_for(count, nthreads, cc_job, &c);
}
