// serde.hip — the reference's on-disk artefacts, read and written by the resident engine.
//
// The reference re-reads two files on EVERY request (halo2-circuits/src/ecc/ecdsa_p256.rs:338-343, 388-393) and
// writes them at :261-270 / in halo2-base `gen_srs`:
//   ./params/kzg_bn254_{k}.srs   ParamsKZG::write / read                       (SRS)
//   ./keys/proving_key.pk        ProvingKey::write / read, SerdeFormat::RawBytes
//   ./keys/verifying_key.vk      VerifyingKey::write / read, SerdeFormat::RawBytes
// so that artefacts interchange with the unchanged Rust host, the engine speaks the
// same byte layouts.  They live in halo2_proofs (PSE fork, not under /root/reference) and are restated here from
// the published code [RECALLED — poly/kzg/commitment.rs `write_custom`, plonk.rs `VerifyingKey::write`,
// `ProvingKey::write`, plonk/permutation.rs, poly.rs `Polynomial::write`, helpers.rs `SerdeFormat`]:
//
//   SerdeFormat      Processed: field elements canonical little-endian (`to_repr`), points compressed (x LE, sign of
//                    y in bit 7 of the last byte, identity = all zero — the encoding the Blake2b transcript also uses);
//                    RawBytes: the in-memory Montgomery limbs, little-endian (Fr 32 B, G1Affine x||y 64 B, G2Affine
//                    x.c0||x.c1||y.c0||y.c1 128 B), validated on read; RawBytesUnchecked: the same bytes, no validation.
//   ParamsKZG        u32 LE k | g[0..n) | g_lagrange[0..n) | g2 | s_g2          (Params::write = RawBytes)
//   Polynomial       u32 BE len | len field elements;   slice of polynomials: u32 BE count | polynomials
//   VerifyingKey     u32 BE k | u32 BE #fixed | fixed commitments | permutation commitments (one per permutation
//                    column, no count) | selectors: per selector 2^k bits packed LSB-first into 2^k / 8 bytes
//                    (transcript_repr is NOT in the file: the Rust host re-hashes the pinned vk after reading — here
//                    it is supplied by the caller, zk_vk_load / zk_pk_read / zk_pk_set_transcript_repr)
//   ProvingKey       vk | l0 | l_last | l_active_row (extended-coset polynomials) | fixed_values | fixed_polys |
//                    fixed_cosets | permutation.permutations | .polys | .cosets   (six polynomial slices)
// Selector order: halo2-lib creates one `q_enable` selector per gate column, then (single-column strategy) the complex
// `q_lookup` selector; selector compression turns every enabled one into its own fixed column (they are not mutually
// exclusive), never-enabled ones into the constant 0 — the fixed-column order constants, table, selectors that the
// reference's k=17 verifying key shows (proving-server/P256Verifier.yul:880-926).
// Nothing of this exists in the reference's tests as bytes, so the formats are parity-unpinned beyond the known answers
// they carry: s_g2 of the SRS (yul:1131-1134) and the k=17 vk commitments (yul:880-980), see tests/.
#include <string.h>

#include "pk.h"
#include "vkrepr.h"

namespace zk {
void launch_to_mont(Fr* a, uint32_t n, hipStream_t st);
}

namespace {

// ------------------------------------------------------------ device side ---
struct Words8 {
    uint32_t w[8];
};

__device__ __forceinline__ bool words_lt_p(const uint32_t* v, const uint32_t* p) {
    for (int i = 7; i >= 0; i--) {
        if (v[i] != p[i]) return v[i] < p[i];
    }
    return false;
}

__device__ __forceinline__ Fq fq_three_mont() {
    Fq t = Fq::zero();
    t.v[0] = 3;
    return fe_to_mont(t);
}

__device__ bool g1_on_curve(const Fq& x, const Fq& y) {
    const Fq rhs = fe_add(fe_mul(fe_sqr(x), x), fq_three_mont());
    return fe_sqr(y) == rhs;
}

// compressed (32 B: x canonical LE, bit 255 = y odd, all-zero = identity) -> affine Montgomery
__global__ void g1_decompress_kernel(const uint8_t* __restrict__ in, G1Affine* __restrict__ out, uint32_t n, Words8 sqrt_exp,
                                     uint32_t* __restrict__ err) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fq x;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(in + (size_t)i * 32);
#pragma unroll
    for (int k = 0; k < 8; k++) x.v[k] = src[k];
    const uint32_t sign = x.v[7] >> 31;
    x.v[7] &= 0x7fffffffu;
    G1Affine r;
    r.x = Fq::zero();
    r.y = Fq::zero();
    if (!words_lt_p(x.v, FqParams::P)) {
        atomicOr(err, 1u);
    } else if (!(x.is_zero() && !sign)) {
        const Fq xm = fe_to_mont(x);
        const Fq rhs = fe_add(fe_mul(fe_sqr(xm), xm), fq_three_mont());
        Fq y = fe_pow(rhs, sqrt_exp.w);  // p = 3 (mod 4): rhs^((p+1)/4)
        if (fe_sqr(y) != rhs) {
            atomicOr(err, 1u);
        } else {
            if ((fe_from_mont(y).v[0] & 1u) != sign) y = fe_neg(y);
            r.x = xm;
            r.y = y;
        }
    }
    fe_store(&out[i].x, r.x);
    fe_store(&out[i].y, r.y);
}

__global__ void g1_compress_kernel(const G1Affine* __restrict__ in, uint8_t* __restrict__ out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const G1Affine p = affine_load(in + i);
    uint32_t* dst = reinterpret_cast<uint32_t*>(out + (size_t)i * 32);
    if (affine_is_identity(p)) {
#pragma unroll
        for (int k = 0; k < 8; k++) dst[k] = 0;
        return;
    }
    Fq x = fe_from_mont(p.x);
    x.v[7] |= (fe_from_mont(p.y).v[0] & 1u) << 31;
#pragma unroll
    for (int k = 0; k < 8; k++) dst[k] = x.v[k];
}

// RawBytes (checked): both coordinates reduced, point on the curve (or the identity (0,0))
__global__ void g1_validate_kernel(const G1Affine* __restrict__ in, uint32_t n, uint32_t* __restrict__ err) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const G1Affine p = affine_load(in + i);
    if (!words_lt_p(p.x.v, FqParams::P) || !words_lt_p(p.y.v, FqParams::P)) {
        atomicOr(err, 1u);
        return;
    }
    if (!affine_is_identity(p) && !g1_on_curve(p.x, p.y)) atomicOr(err, 1u);
}

__global__ void fr_validate_kernel(const Fr* __restrict__ in, size_t n, uint32_t* __restrict__ err) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const Fr v = fe_load(in + i);
    if (!words_lt_p(v.v, FrParams::P)) atomicOr(err, 1u);
}

__global__ void fr_from_mont_kernel(Fr* a, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) fe_store(a + i, fe_from_mont(fe_load(a + i)));
}
__global__ void fr_to_mont_kernel(Fr* a, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) fe_store(a + i, fe_to_mont(fe_load(a + i)));
}

// -------------------------------------------------------------- host side ---
Words8 fq_sqrt_exp() {  // (p + 1) / 4
    Words8 e;
    uint64_t carry = 1;
    uint32_t t[8];
    for (int i = 0; i < 8; i++) {
        const uint64_t s = (uint64_t)FqParams::P[i] + carry;
        t[i] = (uint32_t)s;
        carry = s >> 32;
    }
    for (int i = 0; i < 8; i++) e.w[i] = (t[i] >> 2) | (i + 1 < 8 ? t[i + 1] << 30 : 0);
    return e;
}

bool host_lt_p(const uint32_t* v, const uint32_t* p) {
    for (int i = 7; i >= 0; i--)
        if (v[i] != p[i]) return v[i] < p[i];
    return false;
}

Fq fq_small(uint32_t x) {
    Fq t = Fq::zero();
    t.v[0] = x;
    return fe_to_mont(t);
}

bool host_g1_on_curve(const G1Affine& p) {
    return fe_sqr(p.y) == fe_add(fe_mul(fe_sqr(p.x), p.x), fq_small(3));
}

// one G1 point <-> bytes in `format`; returns bytes consumed / produced, 0 on a malformed point
size_t g1_size(int format) { return format == ZK_SERDE_PROCESSED ? 32 : 64; }

bool host_g1_read(const uint8_t* b, int format, G1Affine* out) {
    if (format == ZK_SERDE_PROCESSED) {
        Fq x;
        memcpy(x.v, b, 32);
        const uint32_t sign = x.v[7] >> 31;
        x.v[7] &= 0x7fffffffu;
        if (!host_lt_p(x.v, FqParams::P)) return false;
        if (x.is_zero() && !sign) {
            out->x = Fq::zero();
            out->y = Fq::zero();
            return true;
        }
        const Fq xm = fe_to_mont(x);
        const Fq rhs = fe_add(fe_mul(fe_sqr(xm), xm), fq_small(3));
        Fq y = fe_pow(rhs, fq_sqrt_exp().w);
        if (fe_sqr(y) != rhs) return false;
        if ((fe_from_mont(y).v[0] & 1u) != sign) y = fe_neg(y);
        out->x = xm;
        out->y = y;
        return true;
    }
    memcpy(out, b, 64);
    if (format == ZK_SERDE_RAW_BYTES) {
        if (!host_lt_p(out->x.v, FqParams::P) || !host_lt_p(out->y.v, FqParams::P)) return false;
        if (!affine_is_identity(*out) && !host_g1_on_curve(*out)) return false;
    }
    return true;
}

void host_g1_write(const G1Affine& p, int format, uint8_t* b) {
    if (format == ZK_SERDE_PROCESSED) {
        if (affine_is_identity(p)) {
            memset(b, 0, 32);
            return;
        }
        Fq x = fe_from_mont(p.x);
        x.v[7] |= (fe_from_mont(p.y).v[0] & 1u) << 31;
        memcpy(b, x.v, 32);
        return;
    }
    memcpy(b, &p, 64);
}

// ---- Fq2 = Fq[u] / (u^2 + 1) and the twist y^2 = x^3 + 3 / (9 + u): only for g2 / s_g2 of the SRS file
struct Fq2 {
    Fq c0, c1;
};
Fq2 f2_add(const Fq2& a, const Fq2& b) { return {fe_add(a.c0, b.c0), fe_add(a.c1, b.c1)}; }
Fq2 f2_sub(const Fq2& a, const Fq2& b) { return {fe_sub(a.c0, b.c0), fe_sub(a.c1, b.c1)}; }
Fq2 f2_mul(const Fq2& a, const Fq2& b) {
    return {fe_sub(fe_mul(a.c0, b.c0), fe_mul(a.c1, b.c1)), fe_add(fe_mul(a.c0, b.c1), fe_mul(a.c1, b.c0))};
}
Fq2 f2_inv(const Fq2& a) {
    const Fq d = fe_inv(fe_add(fe_sqr(a.c0), fe_sqr(a.c1)));
    return {fe_mul(a.c0, d), fe_neg(fe_mul(a.c1, d))};
}
bool f2_is_zero(const Fq2& a) { return a.c0.is_zero() && a.c1.is_zero(); }
bool f2_eq(const Fq2& a, const Fq2& b) { return a.c0 == b.c0 && a.c1 == b.c1; }
Fq2 f2_small(uint32_t a, uint32_t b) { return {fq_small(a), fq_small(b)}; }
Fq2 f2_twist_b() { return f2_mul(f2_small(3, 0), f2_inv(f2_small(9, 1))); }

struct G2A {
    Fq2 x, y;
    bool inf;
};
G2A g2_add(const G2A& a, const G2A& b) {
    if (a.inf) return b;
    if (b.inf) return a;
    Fq2 lam;
    if (f2_eq(a.x, b.x)) {
        if (!f2_eq(a.y, b.y)) return G2A{a.x, a.y, true};
        lam = f2_mul(f2_mul(f2_small(3, 0), f2_mul(a.x, a.x)), f2_inv(f2_add(a.y, a.y)));
    } else {
        lam = f2_mul(f2_sub(b.y, a.y), f2_inv(f2_sub(b.x, a.x)));
    }
    G2A r;
    r.inf = false;
    r.x = f2_sub(f2_sub(f2_mul(lam, lam), a.x), b.x);
    r.y = f2_sub(f2_mul(lam, f2_sub(a.x, r.x)), a.y);
    return r;
}
G2A g2_mul(G2A p, const Fr& k_mont) {
    const Fr k = fe_from_mont(k_mont);
    G2A acc{p.x, p.y, true};
    for (int i = 0; i < 256; i++) {
        if ((k.v[i >> 5] >> (i & 31)) & 1) acc = g2_add(acc, p);
        p = g2_add(p, p);
    }
    return acc;
}
Fq fq_from_hex_words(const uint32_t be[8]) {  // big-endian word order, canonical -> Montgomery
    Fq t;
    for (int i = 0; i < 8; i++) t.v[i] = be[7 - i];
    return fe_to_mont(t);
}
G2A g2_generator() {  // the BN254 G2 generator (reference proving-server/P256Verifier.yul:1125-1128 holds it as x.c1, x.c0, y.c1, y.c0)
    static const uint32_t X0[8] = {0x1800DEEF, 0x121F1E76, 0x426A0066, 0x5E5C4479, 0x674322D4, 0xF75EDADD, 0x46DEBD5C, 0xD992F6ED};
    static const uint32_t X1[8] = {0x198E9393, 0x920D483A, 0x7260BFB7, 0x31FB5D25, 0xF1AA4933, 0x35A9E712, 0x97E485B7, 0xAEF312C2};
    static const uint32_t Y0[8] = {0x12C85EA5, 0xDB8C6DEB, 0x4AAB7180, 0x8DCB408F, 0xE3D1E769, 0x0C43D37B, 0x4CE6CC01, 0x66FA7DAA};
    static const uint32_t Y1[8] = {0x090689D0, 0x585FF075, 0xEC9E99AD, 0x690C3395, 0xBC4B3133, 0x70B38EF3, 0x55ACDADC, 0xD122975B};
    return G2A{{fq_from_hex_words(X0), fq_from_hex_words(X1)}, {fq_from_hex_words(Y0), fq_from_hex_words(Y1)}, false};
}
bool g2_on_curve(const G2A& p) { return f2_eq(f2_mul(p.y, p.y), f2_add(f2_mul(f2_mul(p.x, p.x), p.x), f2_twist_b())); }

// sqrt in Fq2 (complex method); false if `a` is not a square
bool f2_sqrt(const Fq2& a, Fq2* out) {
    if (f2_is_zero(a)) {
        *out = a;
        return true;
    }
    const Words8 e = fq_sqrt_exp();
    auto fq_sqrt = [&](const Fq& v, Fq* r) {
        *r = fe_pow(v, e.w);
        return fe_sqr(*r) == v;
    };
    Fq alpha;
    if (!fq_sqrt(fe_add(fe_sqr(a.c0), fe_sqr(a.c1)), &alpha)) return false;
    const Fq half = fe_inv(fq_small(2));
    Fq delta = fe_mul(fe_add(a.c0, alpha), half), x0;
    if (!fq_sqrt(delta, &x0)) {
        delta = fe_mul(fe_sub(a.c0, alpha), half);
        if (!fq_sqrt(delta, &x0)) return false;
    }
    if (x0.is_zero()) return false;
    const Fq x1 = fe_mul(a.c1, fe_inv(fe_add(x0, x0)));
    *out = Fq2{x0, x1};
    return f2_eq(f2_mul(*out, *out), a);
}

size_t g2_size(int format) { return format == ZK_SERDE_PROCESSED ? 64 : 128; }
// raw image: x.c0 || x.c1 || y.c0 || y.c1 Montgomery
void g2_to_raw(const G2A& p, uint8_t raw[128]) {
    if (p.inf) {
        memset(raw, 0, 128);
        return;
    }
    memcpy(raw, p.x.c0.v, 32);
    memcpy(raw + 32, p.x.c1.v, 32);
    memcpy(raw + 64, p.y.c0.v, 32);
    memcpy(raw + 96, p.y.c1.v, 32);
}
G2A g2_from_raw(const uint8_t raw[128]) {
    G2A p;
    memcpy(p.x.c0.v, raw, 32);
    memcpy(p.x.c1.v, raw + 32, 32);
    memcpy(p.y.c0.v, raw + 64, 32);
    memcpy(p.y.c1.v, raw + 96, 32);
    p.inf = f2_is_zero(p.x) && f2_is_zero(p.y);
    return p;
}
void host_g2_write(const uint8_t raw[128], int format, uint8_t* b) {
    if (format != ZK_SERDE_PROCESSED) {
        memcpy(b, raw, 128);
        return;
    }
    const G2A p = g2_from_raw(raw);
    if (p.inf) {
        memset(b, 0, 64);
        return;
    }
    const Fq x0 = fe_from_mont(p.x.c0), x1 = fe_from_mont(p.x.c1);
    memcpy(b, x0.v, 32);
    memcpy(b + 32, x1.v, 32);
    b[63] |= (uint8_t)((fe_from_mont(p.y.c0).v[0] & 1u) << 7);
}
bool host_g2_read(const uint8_t* b, int format, uint8_t raw[128]) {
    if (format != ZK_SERDE_PROCESSED) {
        memcpy(raw, b, 128);
        if (format == ZK_SERDE_RAW_BYTES) {
            const G2A p = g2_from_raw(raw);
            for (int q = 0; q < 4; q++) {
                uint32_t w[8];
                memcpy(w, raw + 32 * q, 32);
                if (!host_lt_p(w, FqParams::P)) return false;
            }
            if (!p.inf && !g2_on_curve(p)) return false;
        }
        return true;
    }
    Fq x0, x1;
    memcpy(x0.v, b, 32);
    memcpy(x1.v, b + 32, 32);
    const uint32_t sign = x1.v[7] >> 31;
    x1.v[7] &= 0x7fffffffu;
    if (!host_lt_p(x0.v, FqParams::P) || !host_lt_p(x1.v, FqParams::P)) return false;
    G2A p;
    p.inf = false;
    if (x0.is_zero() && x1.is_zero() && !sign) {
        memset(raw, 0, 128);
        return true;
    }
    p.x = Fq2{fe_to_mont(x0), fe_to_mont(x1)};
    if (!f2_sqrt(f2_add(f2_mul(f2_mul(p.x, p.x), p.x), f2_twist_b()), &p.y)) return false;
    if ((fe_from_mont(p.y.c0).v[0] & 1u) != sign) p.y = Fq2{fe_neg(p.y.c0), fe_neg(p.y.c1)};
    g2_to_raw(p, raw);
    return true;
}

void put_be32(uint8_t* b, uint32_t v) {
    b[0] = (uint8_t)(v >> 24);
    b[1] = (uint8_t)(v >> 16);
    b[2] = (uint8_t)(v >> 8);
    b[3] = (uint8_t)v;
}
uint32_t get_be32(const uint8_t* b) { return ((uint32_t)b[0] << 24) | ((uint32_t)b[1] << 16) | ((uint32_t)b[2] << 8) | b[3]; }

bool format_ok(int f) { return f == ZK_SERDE_PROCESSED || f == ZK_SERDE_RAW_BYTES || f == ZK_SERDE_RAW_BYTES_UNCHECKED; }

uint32_t blocks_for(size_t n, uint32_t t = 256) { return (uint32_t)((n + t - 1) / t); }

// a bounded reader / writer over the caller's buffer
struct Out {
    uint8_t* p;
    size_t cap, pos = 0;
    bool real;  // false: size computation only
    Out(uint8_t* buf, size_t c) : p(buf), cap(c), real(buf != nullptr) {}
    uint8_t* take(size_t n) {
        uint8_t* r = (real && pos + n <= cap) ? p + pos : nullptr;
        if (real && pos + n > cap) real = false;  // overflow: keep counting, report ZK_EINVAL at the end
        pos += n;
        return r;
    }
};
struct In {
    const uint8_t* p;
    size_t len, pos = 0;
    const uint8_t* take(size_t n) {
        if (n > len - pos) return nullptr;
        const uint8_t* r = p + pos;
        pos += n;
        return r;
    }
};

}  // namespace

// engine.hip
int srs_alloc(zk_ctx* c, uint32_t k);
int srs_build_tables(zk_ctx* c, uint32_t k);

// s_g2 = [s]G2 after zk_srs_setup (called from engine.hip with the secret)
void srs_set_g2_from_secret(zk_ctx* c, const Fr& s_mont) {
    const G2A g = g2_generator();
    g2_to_raw(g, c->g2_raw);
    g2_to_raw(g2_mul(g, s_mont), c->s_g2_raw);
    c->g2_valid = true;
}

// ================================================================== SRS =====

ZK_API(zk_srs_set_g2, (zk_ctx* c, const uint64_t g2[16], const uint64_t s_g2[16]), (c, g2, s_g2)) {
    if (!c || !g2 || !s_g2) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->srs_k < 0) return ZK_ESTATE;
    uint8_t a[128], b[128];
    if (!host_g2_read((const uint8_t*)g2, ZK_SERDE_RAW_BYTES, a) || !host_g2_read((const uint8_t*)s_g2, ZK_SERDE_RAW_BYTES, b)) return ZK_EINVAL;
    memcpy(c->g2_raw, a, 128);
    memcpy(c->s_g2_raw, b, 128);
    c->g2_valid = true;
    return ZK_OK;
}

ZK_API(zk_srs_write, (zk_ctx* c, int format, uint8_t* out, size_t cap, size_t* len), (c, format, out, cap, len)) {
    if (!c || !len || !format_ok(format)) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    if (c->srs_k < 0 || !c->g2_valid) return ZK_ESTATE;  // after zk_srs_load the G2 half must be given: zk_srs_set_g2
    const size_t n = (size_t)1 << c->srs_k;
    const size_t gs = g1_size(format);
    const size_t total = 4 + 2 * n * gs + 2 * g2_size(format);
    *len = total;
    if (!out) return ZK_OK;
    if (cap < total) return ZK_EINVAL;
    int rc = ctx_bind(c);
    if (rc) return rc;
    const uint32_t k = (uint32_t)c->srs_k;
    out[0] = (uint8_t)k;
    out[1] = (uint8_t)(k >> 8);
    out[2] = (uint8_t)(k >> 16);
    out[3] = (uint8_t)(k >> 24);
    uint8_t* p = out + 4;
    for (int b = 0; b < 2; b++) {
        const G1Affine* src = b ? c->g_lagrange : c->g;
        if (format == ZK_SERDE_PROCESSED) {
            uint8_t* tmp = nullptr;
            if (hipMalloc(&tmp, n * 32) != hipSuccess) return ZK_ENOMEM;
            hipLaunchKernelGGL(g1_compress_kernel, dim3(blocks_for(n)), dim3(256), 0, c->stream, src, tmp, (uint32_t)n);
            hipError_t e = hipMemcpyAsync(p, tmp, n * 32, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            hipFree(tmp);
            if (e != hipSuccess) {
                c->last_hip = (int)e;
                return ZK_EHIP;
            }
        } else {
            HIPCHK(c, hipMemcpy(p, src, n * 64, hipMemcpyDeviceToHost));
        }
        p += n * gs;
    }
    host_g2_write(c->g2_raw, format, p);
    host_g2_write(c->s_g2_raw, format, p + g2_size(format));
    return ZK_OK;
}

ZK_API(zk_srs_read, (zk_ctx* c, const uint8_t* bytes, size_t len, int format), (c, bytes, len, format)) {
    if (!c || !bytes || len < 4 || !format_ok(format)) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    const uint32_t k = (uint32_t)bytes[0] | ((uint32_t)bytes[1] << 8) | ((uint32_t)bytes[2] << 16) | ((uint32_t)bytes[3] << 24);
    if (k < 1 || k > 24) return ZK_EINVAL;
    const size_t n = (size_t)1 << k;
    const size_t gs = g1_size(format);
    if (len != 4 + 2 * n * gs + 2 * g2_size(format)) return ZK_EINVAL;
    uint8_t g2[128], s_g2[128];
    if (!host_g2_read(bytes + 4 + 2 * n * gs, format, g2) || !host_g2_read(bytes + 4 + 2 * n * gs + g2_size(format), format, s_g2)) return ZK_EINVAL;
    int rc = ctx_bind(c);
    if (rc) return rc;
    ctx_release_spares(c);  // the old SRS, its tables and the new bases are alive together below: parked vectors go first
    // decode and validate into fresh buffers: a malformed file leaves the resident SRS, its tables and the keys made under
    // it as they were (they are replaced only once every point has been accepted)
    G1Affine* fresh[2] = {nullptr, nullptr};
    uint32_t* d_err = nullptr;
    uint8_t* tmp = nullptr;
    auto drop = [&]() {
        hipFree(fresh[0]);
        hipFree(fresh[1]);
        hipFree(d_err);
        hipFree(tmp);
    };
    if (hipMalloc(&fresh[0], n * sizeof(G1Affine)) != hipSuccess || hipMalloc(&fresh[1], n * sizeof(G1Affine)) != hipSuccess ||
        hipMalloc(&d_err, 4) != hipSuccess) {
        drop();
        return ZK_ENOMEM;
    }
    hipMemsetAsync(d_err, 0, 4, c->stream);
    hipError_t e = hipSuccess;
    for (int b = 0; b < 2 && e == hipSuccess; b++) {
        G1Affine* dst = fresh[b];
        const uint8_t* src = bytes + 4 + (size_t)b * n * gs;
        if (format == ZK_SERDE_PROCESSED) {
            if (!tmp && hipMalloc(&tmp, n * 32) != hipSuccess) {
                drop();
                return ZK_ENOMEM;
            }
            e = hipMemcpyAsync(tmp, src, n * 32, hipMemcpyHostToDevice, c->stream);
            hipLaunchKernelGGL(g1_decompress_kernel, dim3(blocks_for(n, 64)), dim3(64), 0, c->stream, tmp, dst, (uint32_t)n, fq_sqrt_exp(), d_err);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);  // tmp is reused by the second basis
        } else {
            e = hipMemcpyAsync(dst, src, n * 64, hipMemcpyHostToDevice, c->stream);
            if (format == ZK_SERDE_RAW_BYTES)
                hipLaunchKernelGGL(g1_validate_kernel, dim3(blocks_for(n, 64)), dim3(64), 0, c->stream, dst, (uint32_t)n, d_err);
        }
    }
    uint32_t herr = 0;
    if (e == hipSuccess) e = hipMemcpyAsync(&herr, d_err, 4, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) {
        drop();
        c->last_hip = (int)e;
        return ZK_EHIP;
    }
    if (herr) {  // a point off the curve / a non-canonical coordinate: halo2's read fails the same way
        drop();
        return ZK_EINVAL;
    }
    hipFree(d_err);
    hipFree(tmp);
    srs_adopt(c, k, fresh[0], fresh[1]);
    if ((rc = srs_build_tables(c, k)) != ZK_OK) return rc;
    c->srs_k = (int)k;
    memcpy(c->g2_raw, g2, 128);
    memcpy(c->s_g2_raw, s_g2, 128);
    c->g2_valid = true;
    return ZK_OK;
}

// =========================================================== vk / pk ========

namespace {

uint32_t n_selectors(const Layout& lay) { return lay.n_gate + (lay.single ? 1 : 0); }
// fixed column holding selector s (NO_SELECTOR: compressed away, all-false)
uint32_t selector_column(const Layout& lay, uint32_t s) { return s < lay.n_gate ? lay.fx_sel[s] : lay.fx_qlookup; }

size_t vk_size(const Layout& lay, int format) {
    return 8 + (lay.n_fix + lay.perm_cols.size()) * g1_size(format) + (size_t)n_selectors(lay) * (lay.n / 8);
}
size_t fr_size() { return 32; }
size_t poly_size(size_t len) { return 4 + len * fr_size(); }
size_t pk_size(const Layout& lay, int format) {
    const size_t n = lay.n, N = 4 * n, m = lay.perm_cols.size();
    return vk_size(lay, format) + 3 * poly_size(N) + 3 * 4 + lay.n_fix * (2 * poly_size(n) + poly_size(N)) + 3 * 4 +
           m * (2 * poly_size(n) + poly_size(N));
}

// selector bits of the resident key: the selector columns are 0/1-valued fixed columns
int selector_bits(zk_ctx* c, const zk_pk_rec* pk, std::vector<std::vector<uint8_t>>* out) {
    const Layout& lay = pk->lay;
    const uint32_t n = lay.n;
    std::vector<Fr> col(n);
    out->assign(n_selectors(lay), std::vector<uint8_t>(n / 8, 0));
    for (uint32_t s = 0; s < n_selectors(lay); s++) {
        const uint32_t f = selector_column(lay, s);
        if (f == NO_SELECTOR) continue;
        HIPCHK(c, hipMemcpy(col.data(), pk->fixed_val[f], (size_t)n * sizeof(Fr), hipMemcpyDeviceToHost));
        for (uint32_t r = 0; r < n; r++) {
            if (col[r].is_zero()) continue;
            if (col[r] != Fr::one()) return ZK_ESTATE;  // not a selector column
            (*out)[s][r >> 3] |= (uint8_t)(1u << (r & 7));
        }
    }
    return ZK_OK;
}

int write_vk(zk_ctx* c, const zk_pk_rec* pk, int format, Out& o) {
    const Layout& lay = pk->lay;
    if (uint8_t* p = o.take(4)) put_be32(p, lay.k);
    if (uint8_t* p = o.take(4)) put_be32(p, lay.n_fix);
    for (uint32_t i : vkrepr::halo2_fixed_order(lay))  // halo2's column order: the table column first (vkrepr.h)
        if (uint8_t* p = o.take(g1_size(format))) host_g1_write(pk->fixed_commit[i], format, p);
    for (const G1Affine& cm : pk->perm_commit)
        if (uint8_t* p = o.take(g1_size(format))) host_g1_write(cm, format, p);
    if (!o.real) {
        o.take((size_t)n_selectors(lay) * (lay.n / 8));
        return ZK_OK;
    }
    std::vector<std::vector<uint8_t>> bits;
    int rc = selector_bits(c, pk, &bits);
    if (rc) return rc;
    for (auto& b : bits)
        if (uint8_t* p = o.take(b.size())) memcpy(p, b.data(), b.size());
    return ZK_OK;
}

struct ParsedVk {
    std::vector<G1Affine> fixed, perm;
    std::vector<std::vector<uint8_t>> selectors;
};
int parse_vk(const Layout& lay, In& in, int format, ParsedVk* out) {
    const uint8_t* p = in.take(8);
    if (!p || get_be32(p) != lay.k || get_be32(p + 4) != lay.n_fix) return ZK_EINVAL;
    out->fixed.resize(lay.n_fix);
    out->perm.resize(lay.perm_cols.size());
    for (uint32_t i : vkrepr::halo2_fixed_order(lay)) {  // the file lists the fixed columns in halo2's column order
        const uint8_t* b = in.take(g1_size(format));
        if (!b || !host_g1_read(b, format, &out->fixed[i])) return ZK_EINVAL;
    }
    for (G1Affine& cm : out->perm) {
        const uint8_t* b = in.take(g1_size(format));
        if (!b || !host_g1_read(b, format, &cm)) return ZK_EINVAL;
    }
    out->selectors.assign(n_selectors(lay), std::vector<uint8_t>(lay.n / 8));
    for (auto& s : out->selectors) {
        const uint8_t* b = in.take(s.size());
        if (!b) return ZK_EINVAL;
        memcpy(s.data(), b, s.size());
    }
    return ZK_OK;
}

// one polynomial: u32 BE length, then the values (device -> bytes in `format`)
int write_poly(zk_ctx* c, const Fr* d, size_t len, int format, Out& o, Fr* d_tmp) {
    if (uint8_t* p = o.take(4)) put_be32(p, (uint32_t)len);
    uint8_t* p = o.take(len * fr_size());
    if (!p) return ZK_OK;
    if (format == ZK_SERDE_PROCESSED) {
        HIPCHK(c, hipMemcpyAsync(d_tmp, d, len * sizeof(Fr), hipMemcpyDeviceToDevice, c->stream));
        hipLaunchKernelGGL(fr_from_mont_kernel, dim3(blocks_for(len)), dim3(256), 0, c->stream, d_tmp, len);
        d = d_tmp;
    }
    HIPCHK(c, hipMemcpyAsync(p, d, len * sizeof(Fr), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return ZK_OK;
}
int read_poly(zk_ctx* c, In& in, size_t len, int format, Fr* d, uint32_t* d_err) {
    const uint8_t* h = in.take(4);
    if (!h || get_be32(h) != len) return ZK_EINVAL;
    const uint8_t* p = in.take(len * fr_size());
    if (!p) return ZK_EINVAL;
    HIPCHK(c, hipMemcpyAsync(d, p, len * sizeof(Fr), hipMemcpyHostToDevice, c->stream));
    if (format != ZK_SERDE_RAW_BYTES_UNCHECKED)
        hipLaunchKernelGGL(fr_validate_kernel, dim3(blocks_for(len)), dim3(256), 0, c->stream, d, len, d_err);
    if (format == ZK_SERDE_PROCESSED) hipLaunchKernelGGL(fr_to_mont_kernel, dim3(blocks_for(len)), dim3(256), 0, c->stream, d, len);
    return ZK_OK;
}

}  // namespace

ZK_API(zk_vk_write, (zk_ctx* c, zk_pk h, int format, uint8_t* out, size_t cap, size_t* len), (c, h, format, out, cap, len)) {
    if (!c || !len || !format_ok(format)) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    auto it = c->pks.find(h);
    if (it == c->pks.end()) return ZK_EINVAL;
    if (it->second->srs_gen != c->srs_gen) return ZK_ESTATE;
    *len = vk_size(it->second->lay, format);
    if (!out) return ZK_OK;
    if (cap < *len) return ZK_EINVAL;
    int rc = ctx_bind(c);
    if (rc) return rc;
    Out o(out, cap);
    return write_vk(c, it->second, format, o);
}

// adopt the Rust host's VerifyingKey for a resident key: its commitments and selectors must be the key's own
// (ZK_EINVAL otherwise — a vk of another circuit or another SRS), and the host's transcript_repr replaces the stand-in
ZK_API(zk_vk_load, (zk_ctx* c, zk_pk h, const uint8_t* bytes, size_t len, int format, const uint64_t transcript_repr[4]), (c, h, bytes, len, format, transcript_repr)) {
    if (!c || !bytes || !format_ok(format)) return ZK_EINVAL;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        auto it = c->pks.find(h);
        if (it == c->pks.end()) return ZK_EINVAL;
        zk_pk_rec* pk = it->second;
        if (pk->srs_gen != c->srs_gen) return ZK_ESTATE;
        if (len != vk_size(pk->lay, format)) return ZK_EINVAL;
        In in{bytes, len};
        ParsedVk vk;
        int rc = parse_vk(pk->lay, in, format, &vk);
        if (rc) return rc;
        for (size_t i = 0; i < vk.fixed.size(); i++)
            if (memcmp(&vk.fixed[i], &pk->fixed_commit[i], sizeof(G1Affine)) != 0) return ZK_EINVAL;
        for (size_t i = 0; i < vk.perm.size(); i++)
            if (memcmp(&vk.perm[i], &pk->perm_commit[i], sizeof(G1Affine)) != 0) return ZK_EINVAL;
        if ((rc = ctx_bind(c))) return rc;
        std::vector<std::vector<uint8_t>> bits;
        if ((rc = selector_bits(c, pk, &bits))) return rc;
        if (bits != vk.selectors) return ZK_EINVAL;
    }
    return transcript_repr ? zk_pk_set_transcript_repr(c, h, transcript_repr) : ZK_OK;
}

ZK_API(zk_pk_write, (zk_ctx* c, zk_pk h, int format, uint8_t* out, size_t cap, size_t* len), (c, h, format, out, cap, len)) {
    if (!c || !len || !format_ok(format)) return ZK_EINVAL;
    std::lock_guard<std::mutex> lk(c->mu);
    auto it = c->pks.find(h);
    if (it == c->pks.end()) return ZK_EINVAL;
    zk_pk_rec* pk = it->second;
    if (pk->srs_gen != c->srs_gen) return ZK_ESTATE;
    const Layout& lay = pk->lay;
    *len = pk_size(lay, format);
    if (!out) return ZK_OK;
    if (cap < *len) return ZK_EINVAL;
    int rc = ctx_bind(c);
    if (rc) return rc;
    const size_t n = lay.n, N = 4 * n;
    Out o(out, cap);
    if ((rc = write_vk(c, pk, format, o))) return rc;
    Fr* tmp = pk->h_ext;  // 4n elements of workspace, free between proofs
    if ((rc = write_poly(c, pk->l0_coset, N, format, o, tmp)) || (rc = write_poly(c, pk->l_last_coset, N, format, o, tmp)) ||
        (rc = write_poly(c, pk->l_active_coset, N, format, o, tmp)))
        return rc;
    // fixed columns in halo2's column order (table first), permutation columns as they are
    const std::vector<uint32_t> forder = vkrepr::halo2_fixed_order(lay);
    std::vector<uint32_t> sorder(lay.perm_cols.size());
    for (uint32_t i = 0; i < sorder.size(); i++) sorder[i] = i;
    auto slice = [&](const std::vector<Fr*>& v, const std::vector<uint32_t>& order, size_t len_each) -> int {
        if (uint8_t* p = o.take(4)) put_be32(p, (uint32_t)v.size());
        for (uint32_t i : order)
            if (int r = write_poly(c, v[i], len_each, format, o, tmp)) return r;
        return ZK_OK;
    };
    if ((rc = slice(pk->fixed_val, forder, n)) || (rc = slice(pk->fixed_poly, forder, n)) || (rc = slice(pk->fixed_coset, forder, N)) ||
        (rc = slice(pk->sigma_val, sorder, n)) || (rc = slice(pk->sigma_poly, sorder, n)) || (rc = slice(pk->sigma_coset, sorder, N)))
        return rc;
    return o.pos == *len && o.real ? ZK_OK : ZK_EINTERNAL;
}

ZK_API(zk_pk_read, (zk_ctx* c, const zk_circuit_params* params, const uint8_t* bytes, size_t len, int format, const uint64_t transcript_repr[4], zk_pk* out), (c, params, bytes, len, format, transcript_repr, out)) {
    if (!c || !params || !bytes || !out || !format_ok(format)) return ZK_EINVAL;
    uint64_t handle = 0;
    {
        std::lock_guard<std::mutex> lk(c->mu);
        int rc = ctx_bind(c);
        if (rc) return rc;
        ctx_release_spares(c);
        Layout lay;
        if (params->num_advice > 1 && 2 * (uint64_t)params->num_idle_gate_columns > params->num_advice) return ZK_ELAYOUT;
        if (!lay.init(*params)) return ZK_EINVAL;
        if (c->srs_k != (int)lay.k) return ZK_ESTATE;
        if (len != pk_size(lay, format)) return ZK_EINVAL;
        const size_t n = lay.n, N = 4 * n, m = lay.perm_cols.size();
        In in{bytes, len};
        ParsedVk vk;
        if ((rc = parse_vk(lay, in, format, &vk))) return rc;
        zk_pk_rec* pk = new (std::nothrow) zk_pk_rec();
        if (!pk) return ZK_ENOMEM;
        pk->lay = lay;
        pk->srs_gen = c->srs_gen;
        pk->max_evals = (uint32_t)(lay.advice_queries.size() + lay.n_fix + lay.perm_cols.size() + 3 * lay.n_chunks + 5 * lay.n_lookups + 16);
        pk->fixed_commit = vk.fixed;
        pk->perm_commit = vk.perm;
        Dev d{c, pk};
        uint32_t* d_err = nullptr;
        auto fail = [&](int code) {
            hipStreamSynchronize(c->stream);
            hipFree(d_err);
            pk_destroy(pk);
            return code;
        };
        if (hipMalloc(&d_err, 4) != hipSuccess) return fail(ZK_ENOMEM);
        hipMemsetAsync(d_err, 0, 4, c->stream);
        pk->l0_coset = d.alloc(N);
        pk->l_last_coset = d.alloc(N);
        pk->l_active_coset = d.alloc(N);
        if (d.rc) return fail(d.rc);
        if ((rc = read_poly(c, in, N, format, pk->l0_coset, d_err)) || (rc = read_poly(c, in, N, format, pk->l_last_coset, d_err)) ||
            (rc = read_poly(c, in, N, format, pk->l_active_coset, d_err)))
            return fail(rc);
        // the file lists the fixed columns in halo2's column order (table first): column `pos` of a slice is the
        // engine's column order[pos]
        const std::vector<uint32_t> forder = vkrepr::halo2_fixed_order(lay);
        std::vector<uint32_t> sorder(m);
        for (uint32_t i = 0; i < m; i++) sorder[i] = i;
        auto slice = [&](std::vector<Fr*>& v, const std::vector<uint32_t>& order, size_t len_each) -> int {
            const uint8_t* hcount = in.take(4);
            if (!hcount || get_be32(hcount) != order.size()) return ZK_EINVAL;
            v.assign(order.size(), nullptr);
            for (uint32_t i : order) {
                Fr* p = d.alloc(len_each);
                if (d.rc) return d.rc;
                v[i] = p;
                if (int r = read_poly(c, in, len_each, format, p, d_err)) return r;
            }
            return ZK_OK;
        };
        if ((rc = slice(pk->fixed_val, forder, n)) || (rc = slice(pk->fixed_poly, forder, n)) || (rc = slice(pk->fixed_coset, forder, N)) ||
            (rc = slice(pk->sigma_val, sorder, n)) || (rc = slice(pk->sigma_poly, sorder, n)) || (rc = slice(pk->sigma_coset, sorder, N)))
            return fail(rc);
        if (in.pos != len) return fail(ZK_EINVAL);
        uint32_t herr = 0;
        if (hipMemcpyAsync(&herr, d_err, 4, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess)
            return fail(ZK_EHIP);
        if (herr) return fail(ZK_EINVAL);  // a field element >= r
        // the lookup path of this engine is specialised to halo2-lib's range table; the selector bits of the vk must be
        // the selector columns of the key
        {
            const uint32_t T = 1u << lay.lookup_bits;
            std::vector<Fr> tab(n);
            if (hipMemcpy(tab.data(), pk->fixed_val[lay.fx_table], n * sizeof(Fr), hipMemcpyDeviceToHost) != hipSuccess) return fail(ZK_EHIP);
            for (uint32_t r = 0; r < n; r++) {
                const Fr v = fe_from_mont(tab[r]);
                const uint32_t want = r < T ? r : 0;
                if (v.v[0] != want || v.v[1] | v.v[2] | v.v[3] | v.v[4] | v.v[5] | v.v[6] | v.v[7]) return fail(ZK_EINVAL);
            }
            std::vector<std::vector<uint8_t>> bits;
            if ((rc = selector_bits(c, pk, &bits))) return fail(rc == ZK_ESTATE ? ZK_EINVAL : rc);
            if (bits != vk.selectors) return fail(ZK_EINVAL);
            if (!layout_selectors_fit(lay, bits)) return fail(ZK_ELAYOUT);  // halo2 would have compressed these selectors differently
        }
        hipFree(d_err);
        d_err = nullptr;
        // the file's commitments must belong to the RESIDENT SRS: a key written under another SRS would be stamped with
        // this context's SRS generation and yield proofs that do not verify.  Spot check: the table column (never zero)
        // recommitted in both of its forms
        {
            G1Jac j1, j2;
            if ((rc = ctx_msm_device(c, pk->fixed_val[lay.fx_table], c->g_lagrange, n, &j1)) ||
                (rc = ctx_msm_device(c, pk->fixed_poly[lay.fx_table], c->g, n, &j2)))
                return fail(rc);
            const G1Affine a1 = g1_jac_to_affine_host(j1), a2 = g1_jac_to_affine_host(j2);
            if (memcmp(&a1, &pk->fixed_commit[lay.fx_table], sizeof(G1Affine)) != 0 || memcmp(&a2, &a1, sizeof(G1Affine)) != 0)
                return fail(ZK_EINVAL);
        }
        pk->transcript_repr = pk_standin_transcript_repr(pk);
        if ((rc = pk_alloc_workspace(c, pk))) return fail(rc);
        if (hipStreamSynchronize(c->stream) != hipSuccess || hipGetLastError() != hipSuccess) return fail(ZK_EHIP);
        handle = c->next_handle++;
        c->pks[handle] = pk;
    }
    if (transcript_repr) {
        int rc = zk_pk_set_transcript_repr(c, handle, transcript_repr);
        if (rc) {
            zk_pk_free(c, handle);
            return rc;
        }
    }
    *out = handle;
    return ZK_OK;
}
