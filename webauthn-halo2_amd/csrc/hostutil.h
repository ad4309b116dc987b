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

inline Fr fr_omega(uint32_t k) {  // primitive 2^k-th root: ROOT^(2^(28-k))
    Fr w = fr_root_of_unity_2_28();
    for (uint32_t i = k; i < 28; i++) w = fe_sqr(w);
    return w;
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
