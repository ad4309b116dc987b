// hostutil.h — host-side helpers of libzkmi355.so: field constants, canonical
// <-> Montgomery conversion, ChaCha20 keystream (rand_chacha's ChaCha20Rng, the
// RNG `gen_srs` seeds with [0;32] — reference halo2-circuits/src/ecc/ecdsa_p256.rs:258).
#pragma once
#include <stdint.h>
#include <string.h>

#include "ec.hip.h"
#include "field.hip.h"

namespace zk {

// 512-bit little-endian integer -> Fr (Montgomery), = halo2curves Fr::from_u512
inline Fr fr_from_u512_le(const uint8_t b[64]) {
    // value = lo + hi * 2^256 ; to Montgomery: lo*R + hi*R*2^256 = mul(lo, R2) + mul(hi, R3)
    Fr lo, hi;
    memcpy(lo.v, b, 32);
    memcpy(hi.v, b + 32, 32);
    // lo, hi may exceed p (they are < 2^256): fe_mul's CIOS tolerates one operand < 2^256
    // as long as the other is < p, producing a result < 2p that reduce_once fixes.
    const Fr r2 = Fr::r2();
    const Fr r3 = fe_mul(r2, r2);  // R^2 * R^2 / R = R^3
    return fe_add(fe_mul(r2, lo), fe_mul(r3, hi));
}

inline Fr fr_from_u64(uint64_t x) {
    Fr a = Fr::zero();
    a.v[0] = (uint32_t)x;
    a.v[1] = (uint32_t)(x >> 32);
    return fe_to_mont(a);
}

template <class PRM>
inline Fe<PRM> fe_pow_u64(Fe<PRM> a, uint64_t e) {
    Fe<PRM> acc = Fe<PRM>::one();
    while (e) {
        if (e & 1) acc = fe_mul(acc, a);
        a = fe_sqr(a);
        e >>= 1;
    }
    return acc;
}

// multiplicative generator 7, 2-adicity 28 (halo2curves bn256::Fr)
inline Fr fr_root_of_unity_2_28() {
    // exponent (r - 1) >> 28
    uint32_t rm1[8];
    for (int i = 0; i < 8; i++) rm1[i] = FrParams::P[i];
    rm1[0] -= 1;
    uint32_t t[8];
    for (int i = 0; i < 8; i++) {
        uint64_t lo = rm1[i] >> 28;
        uint64_t hi = (i + 1 < 8) ? ((uint64_t)rm1[i + 1] << 4) : 0;
        t[i] = (uint32_t)(lo | hi);
    }
    return fe_pow(fr_from_u64(7), t);
}

// (the table is made once per process: the root alone is a 226-bit exponentiation, 19 us of host time that used to sit in front
// of every proof and every seam call)
inline Fr fr_omega(uint32_t k) {  // primitive 2^k-th root: ROOT^(2^(28-k))
    struct Table {
        Fr w[29];
        Table() {
            w[28] = fr_root_of_unity_2_28();
            for (int i = 27; i >= 0; i--) w[i] = fe_sqr(w[i + 1]);
        }
    };
    static const Table t;  // (thread-safe since C++11)
    return t.w[k <= 28 ? k : 28];
}

// ---- host-side inversion: binary extended Euclid on 4 x 64-bit limbs ----------------------------------------------------------
// fe_inv (field.hip.h) is Fermat's a^(p-2): ~ 380 Montgomery products, ~ 19 us on a host core — and a lone proof makes some twenty
// of them between its GPU phases (one per MSM pass for the affine forms, the grand products' batch inverse, SHPLONK's Lagrange
// bases), the GPU idle meanwhile.  The inverse of an element is unique, so any algorithm gives the same bytes; this one takes
// ~ 3 us.  Host only (data-dependent loops).
namespace hostinv {
struct U256 {
    uint64_t w[4];
};
inline bool is_one(const U256& a) { return a.w[0] == 1 && !(a.w[1] | a.w[2] | a.w[3]); }
inline bool is_zero(const U256& a) { return !(a.w[0] | a.w[1] | a.w[2] | a.w[3]); }
inline bool ge(const U256& a, const U256& b) {
    for (int i = 3; i >= 0; i--)
        if (a.w[i] != b.w[i]) return a.w[i] > b.w[i];
    return true;
}
inline uint64_t add(U256& a, const U256& b) {  // a += b, returns the carry out
    unsigned __int128 c = 0;
    for (int i = 0; i < 4; i++) {
        c += (unsigned __int128)a.w[i] + b.w[i];
        a.w[i] = (uint64_t)c;
        c >>= 64;
    }
    return (uint64_t)c;
}
inline void sub(U256& a, const U256& b) {  // a -= b (a >= b)
    uint64_t borrow = 0;
    for (int i = 0; i < 4; i++) {
        const uint64_t bi = b.w[i], t = a.w[i] - bi, r = t - borrow;
        borrow = (a.w[i] < bi) | (t < borrow);
        a.w[i] = r;
    }
}
inline void shr1(U256& a, uint64_t top) {  // (top : a) >> 1
    for (int i = 0; i < 3; i++) a.w[i] = (a.w[i] >> 1) | (a.w[i + 1] << 63);
    a.w[3] = (a.w[3] >> 1) | (top << 63);
}
inline void halve_mod(U256& x, const U256& p) {  // x / 2 mod p (p odd, x < p)
    if (x.w[0] & 1) {
        const uint64_t c = add(x, p);
        shr1(x, c);
    } else {
        shr1(x, 0);
    }
}
inline void sub_mod(U256& a, const U256& b, const U256& p) {  // a = a - b mod p (a, b < p)
    if (ge(a, b)) {
        sub(a, b);
    } else {
        add(a, p);  // (no carry out of 256 bits: a + p - b < 2p < 2^255)
        sub(a, b);
    }
}
}  // namespace hostinv

// the inverse of a Montgomery image a R, as a Montgomery image (a^-1 R); 0 -> 0 like fe_inv
template <class PRM>
inline Fe<PRM> fe_inv_fast(const Fe<PRM>& a) {
    using namespace hostinv;
    U256 p, u, v, x1 = {{1, 0, 0, 0}}, x2 = {{0, 0, 0, 0}};
    for (int i = 0; i < 4; i++) {
        p.w[i] = (uint64_t)PRM::P[2 * i] | ((uint64_t)PRM::P[2 * i + 1] << 32);
        u.w[i] = (uint64_t)a.v[2 * i] | ((uint64_t)a.v[2 * i + 1] << 32);
    }
    if (is_zero(u)) return Fe<PRM>::zero();
    v = p;
    while (!is_one(u) && !is_one(v)) {
        while (!(u.w[0] & 1)) {
            shr1(u, 0);
            halve_mod(x1, p);
        }
        while (!(v.w[0] & 1)) {
            shr1(v, 0);
            halve_mod(x2, p);
        }
        if (ge(u, v)) {
            sub(u, v);
            sub_mod(x1, x2, p);
        } else {
            sub(v, u);
            sub_mod(x2, x1, p);
        }
    }
    const U256& r = is_one(u) ? x1 : x2;  // (a R)^-1 = a^-1 R^-1 as a plain integer
    Fe<PRM> t;
    for (int i = 0; i < 4; i++) {
        t.v[2 * i] = (uint32_t)r.w[i];
        t.v[2 * i + 1] = (uint32_t)(r.w[i] >> 32);
    }
    const Fe<PRM> r2 = Fe<PRM>::r2();
    return fe_mul(t, fe_mul(r2, r2));  // x R^3 / R: a^-1 R^-1 R^2 = a^-1 R
}

// halo2curves bn256 Fr::ZETA = 0x30644e72e131a029048b6e193fd84104cc37a73fec2bc5e9b8ca0b2d36636f23 [RECALLED constant; it is a
// primitive cube root of unity, namely (7^((r-1)/3))^2], the extended-domain coset generator `g_coset` of halo2's
// EvaluationDomain.  h(X) — and so every proof byte — is the same for either cube root; the extended cosets inside a
// ProvingKey file (zk_pk_read / zk_pk_write) are not, hence halo2's choice.
inline Fr fr_zeta_root() {  // 7^((r-1)/3)
    // (r-1)/3
    uint32_t rm1[8];
    for (int i = 0; i < 8; i++) rm1[i] = FrParams::P[i];
    rm1[0] -= 1;
    uint32_t q[8];
    uint64_t rem = 0;
    for (int i = 7; i >= 0; i--) {
        uint64_t cur = (rem << 32) | rm1[i];
        q[i] = (uint32_t)(cur / 3);
        rem = cur % 3;
    }
    return fe_pow(fr_from_u64(7), q);
}
inline Fr fr_zeta() { return fe_sqr(fr_zeta_root()); }

// ---- ChaCha20 ---------------------------------------------------------------
inline uint32_t rotl32(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
#define ZK_QR(a, b, c, d)                     \
    a += b; d ^= a; d = rotl32(d, 16);        \
    c += d; b ^= c; b = rotl32(b, 12);        \
    a += b; d ^= a; d = rotl32(d, 8);         \
    c += d; b ^= c; b = rotl32(b, 7);

inline void chacha20_block(const uint8_t key[32], uint64_t counter, uint8_t out[64]) {
    uint32_t st[16] = {0x61707865, 0x3320646e, 0x79622d32, 0x6b206574};
    memcpy(st + 4, key, 32);
    st[12] = (uint32_t)counter;
    st[13] = (uint32_t)(counter >> 32);
    st[14] = 0;
    st[15] = 0;
    uint32_t w[16];
    memcpy(w, st, 64);
    for (int i = 0; i < 10; i++) {
        ZK_QR(w[0], w[4], w[8], w[12]) ZK_QR(w[1], w[5], w[9], w[13]) ZK_QR(w[2], w[6], w[10], w[14]) ZK_QR(w[3], w[7], w[11], w[15])
        ZK_QR(w[0], w[5], w[10], w[15]) ZK_QR(w[1], w[6], w[11], w[12]) ZK_QR(w[2], w[7], w[8], w[13]) ZK_QR(w[3], w[4], w[9], w[14])
    }
    for (int i = 0; i < 16; i++) w[i] += st[i];
    memcpy(out, w, 64);
}

struct ChaCha20Rng {
    uint8_t key[32];
    uint64_t block = 0;
    uint8_t buf[64];
    int pos = 64;
    explicit ChaCha20Rng(const uint8_t seed[32]) { memcpy(key, seed, 32); }
    void fill(uint8_t* out, size_t n) {
        while (n) {
            if (pos == 64) {
                chacha20_block(key, block++, buf);
                pos = 0;
            }
            size_t take = (size_t)(64 - pos) < n ? (size_t)(64 - pos) : n;
            memcpy(out, buf + pos, take);
            pos += (int)take;
            out += take;
            n -= take;
        }
    }
    Fr next_fr() {  // halo2curves Fr::random
        uint8_t b[64];
        fill(b, 64);
        return fr_from_u512_le(b);
    }
};

}  // namespace zk
