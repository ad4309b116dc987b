// transcript.h — host-side Fiat-Shamir transcripts of the prover.
//
// Blake2bWrite<_, G1Affine, Challenge255<_>> (halo2_proofs transcript.rs; chosen by the
// reference at halo2-circuits/src/ecc/ecdsa_p256.rs:415) and snark-verifier's
// EvmTranscript (Keccak-256; reference ecdsa_p256.rs:365,371 — absorb/squeeze rule
// pinned by proving-server/P256Verifier.yul:34,75-81,97-109).  Transcripts stay on
// the host: they hash a few hundred bytes per proof.
#pragma once
#include <stdint.h>
#include <string.h>

#include <vector>

#include "ec.hip.h"
#include "field.hip.h"
#include "hostutil.h"

namespace zk {

// ---------------------------------------------------------------- Keccak ---
inline uint64_t rotl64(uint64_t x, int n) { return n ? (x << n) | (x >> (64 - n)) : x; }

inline void keccak_f1600(uint64_t s[25]) {
    static const uint64_t RC[24] = {
        0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808AULL, 0x8000000080008000ULL,
        0x000000000000808BULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
        0x000000000000008AULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000AULL,
        0x000000008000808BULL, 0x800000000000008BULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
        0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800AULL, 0x800000008000000AULL,
        0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
    static const int ROT[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};
    for (int round = 0; round < 24; round++) {
        uint64_t c[5], d[5], b[25];
        for (int x = 0; x < 5; x++) c[x] = s[x] ^ s[x + 5] ^ s[x + 10] ^ s[x + 15] ^ s[x + 20];
        for (int x = 0; x < 5; x++) d[x] = c[(x + 4) % 5] ^ rotl64(c[(x + 1) % 5], 1);
        for (int i = 0; i < 25; i++) s[i] ^= d[i % 5];
        for (int x = 0; x < 5; x++)
            for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = rotl64(s[x + 5 * y], ROT[x + 5 * y]);
        for (int y = 0; y < 5; y++)
            for (int x = 0; x < 5; x++) s[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);
        s[0] ^= RC[round];
    }
}

inline void keccak256(const uint8_t* data, size_t len, uint8_t out[32]) {
    uint64_t s[25];
    memset(s, 0, sizeof(s));
    const size_t rate = 136;
    std::vector<uint8_t> m(data, data + len);
    m.push_back(0x01);
    while (m.size() % rate) m.push_back(0);
    m.back() |= 0x80;
    for (size_t off = 0; off < m.size(); off += rate) {
        for (size_t i = 0; i < rate / 8; i++) {
            uint64_t w;
            memcpy(&w, &m[off + 8 * i], 8);
            s[i] ^= w;
        }
        keccak_f1600(s);
    }
    memcpy(out, s, 32);
}

// --------------------------------------------------------------- Blake2b ---
struct Blake2b {
    uint64_t h[8];
    uint64_t t0 = 0, t1 = 0;
    uint8_t buf[128];
    size_t buflen = 0;

    static const uint64_t* iv() {
        static const uint64_t IV[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL,
                                       0xa54ff53a5f1d36f1ULL, 0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL,
                                       0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
        return IV;
    }
    // digest length 64, no key, 16-byte personalisation
    explicit Blake2b(const char personal[16]) {
        uint8_t p[64];
        memset(p, 0, 64);
        p[0] = 64;  // digest length
        p[2] = 1;   // fanout
        p[3] = 1;   // depth
        memcpy(p + 48, personal, 16);
        for (int i = 0; i < 8; i++) {
            uint64_t w;
            memcpy(&w, p + 8 * i, 8);
            h[i] = iv()[i] ^ w;
        }
    }
    static inline uint64_t rotr(uint64_t x, int n) { return (x >> n) | (x << (64 - n)); }
    void compress(const uint8_t block[128], bool last) {
        static const uint8_t SIG[12][16] = {
            {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3},
            {11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4}, {7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8},
            {9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13}, {2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9},
            {12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11}, {13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10},
            {6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5}, {10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0},
            {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15}, {14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3}};
        uint64_t m[16], v[16];
        memcpy(m, block, 128);
        for (int i = 0; i < 8; i++) {
            v[i] = h[i];
            v[i + 8] = iv()[i];
        }
        v[12] ^= t0;
        v[13] ^= t1;
        if (last) v[14] = ~v[14];
#define ZK_B2G(a, b, c, d, x, y)       \
    v[a] = v[a] + v[b] + (x);          \
    v[d] = rotr(v[d] ^ v[a], 32);      \
    v[c] = v[c] + v[d];                \
    v[b] = rotr(v[b] ^ v[c], 24);      \
    v[a] = v[a] + v[b] + (y);          \
    v[d] = rotr(v[d] ^ v[a], 16);      \
    v[c] = v[c] + v[d];                \
    v[b] = rotr(v[b] ^ v[c], 63);
        for (int r = 0; r < 12; r++) {
            const uint8_t* s = SIG[r];
            ZK_B2G(0, 4, 8, 12, m[s[0]], m[s[1]])
            ZK_B2G(1, 5, 9, 13, m[s[2]], m[s[3]])
            ZK_B2G(2, 6, 10, 14, m[s[4]], m[s[5]])
            ZK_B2G(3, 7, 11, 15, m[s[6]], m[s[7]])
            ZK_B2G(0, 5, 10, 15, m[s[8]], m[s[9]])
            ZK_B2G(1, 6, 11, 12, m[s[10]], m[s[11]])
            ZK_B2G(2, 7, 8, 13, m[s[12]], m[s[13]])
            ZK_B2G(3, 4, 9, 14, m[s[14]], m[s[15]])
        }
#undef ZK_B2G
        for (int i = 0; i < 8; i++) h[i] ^= v[i] ^ v[i + 8];
    }
    void update(const uint8_t* in, size_t len) {
        while (len) {
            if (buflen == 128) {  // buffer full and more input follows: not the last block
                t0 += 128;
                if (t0 < 128) t1++;
                compress(buf, false);
                buflen = 0;
            }
            size_t take = 128 - buflen < len ? 128 - buflen : len;
            memcpy(buf + buflen, in, take);
            buflen += take;
            in += take;
            len -= take;
        }
    }
    // finalize a COPY of the state (the transcript keeps absorbing afterwards)
    void finalize_copy(uint8_t out[64]) const {
        Blake2b c = *this;
        c.t0 += c.buflen;
        if (c.t0 < c.buflen) c.t1++;
        memset(c.buf + c.buflen, 0, 128 - c.buflen);
        c.compress(c.buf, true);
        memcpy(out, c.h, 64);
    }
};

// ------------------------------------------------------------ transcripts ---
inline void fe_to_be32(const uint32_t v[8], uint8_t out[32]) {
    for (int i = 0; i < 8; i++) {
        uint32_t w = v[7 - i];
        out[4 * i] = (uint8_t)(w >> 24);
        out[4 * i + 1] = (uint8_t)(w >> 16);
        out[4 * i + 2] = (uint8_t)(w >> 8);
        out[4 * i + 3] = (uint8_t)w;
    }
}

struct Transcript {
    virtual ~Transcript() {}
    virtual void common_scalar(const Fr& s_mont) = 0;
    virtual void write_scalar(const Fr& s_mont) = 0;
    virtual bool write_point(const G1Affine& p_mont) = 0;  // false for the identity
    virtual Fr squeeze() = 0;                             // Montgomery
    std::vector<uint8_t> out;
};

struct EvmTranscript : Transcript {
    std::vector<uint8_t> buf;
    void common_scalar(const Fr& s) override {
        uint8_t b[32];
        fe_to_be32(fe_from_mont(s).v, b);
        buf.insert(buf.end(), b, b + 32);
    }
    void write_scalar(const Fr& s) override {
        uint8_t b[32];
        fe_to_be32(fe_from_mont(s).v, b);
        buf.insert(buf.end(), b, b + 32);
        out.insert(out.end(), b, b + 32);
    }
    bool write_point(const G1Affine& p) override {
        if (affine_is_identity(p)) return false;
        uint8_t b[64];
        fe_to_be32(fe_from_mont(p.x).v, b);
        fe_to_be32(fe_from_mont(p.y).v, b + 32);
        buf.insert(buf.end(), b, b + 64);
        out.insert(out.end(), b, b + 64);
        return true;
    }
    Fr squeeze() override {
        if (buf.size() == 32) buf.push_back(0x01);
        uint8_t h[32];
        keccak256(buf.data(), buf.size(), h);
        buf.assign(h, h + 32);
        // challenge = hash (big-endian integer) mod r, to Montgomery
        uint8_t le[64];
        memset(le, 0, 64);
        for (int i = 0; i < 32; i++) le[i] = h[31 - i];
        return fr_from_u512_le(le);
    }
};

struct Blake2bTranscript : Transcript {
    Blake2b st;
    Blake2bTranscript() : st("Halo2-Transcript") {}
    void absorb_scalar(const Fr& s, uint8_t le[32]) {
        const Fr c = fe_from_mont(s);
        memcpy(le, c.v, 32);
        const uint8_t pre = 2;
        st.update(&pre, 1);
        st.update(le, 32);
    }
    void common_scalar(const Fr& s) override {
        uint8_t le[32];
        absorb_scalar(s, le);
    }
    void write_scalar(const Fr& s) override {
        uint8_t le[32];
        absorb_scalar(s, le);
        out.insert(out.end(), le, le + 32);
    }
    bool write_point(const G1Affine& p) override {
        if (affine_is_identity(p)) return false;
        const Fq x = fe_from_mont(p.x), y = fe_from_mont(p.y);
        const uint8_t pre = 1;
        st.update(&pre, 1);
        st.update((const uint8_t*)x.v, 32);
        st.update((const uint8_t*)y.v, 32);
        uint8_t c[32];
        memcpy(c, x.v, 32);
        c[31] |= (uint8_t)((y.v[0] & 1) << 7);
        out.insert(out.end(), c, c + 32);
        return true;
    }
    Fr squeeze() override {
        const uint8_t pre = 0;
        st.update(&pre, 1);
        uint8_t h[64];
        st.finalize_copy(h);
        return fr_from_u512_le(h);
    }
};

}  // namespace zk
