// field.hip.h — BN254 Fr / Fq arithmetic for gfx950 device code.
//
// Device twin of halo2curves `bn256::{Fr,Fq}` (4 x u64-limb Montgomery, R = 2^256;
// reached from the reference through halo2_proofs at
// halo2-circuits/src/ecc/ecdsa_p256.rs:366-373,416-423 — SURVEY.md §8a a10).
// CDNA4 has no 64-bit integer multiplier: an element is 8 x 32-bit limbs in
// VGPRs (same little-endian memory image as 4 x u64) and products go through
// v_mad_u64_u32 (32x32+64 -> 64).  Values are kept fully reduced ([0, p)).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace zk {

struct FrParams {
    static constexpr uint32_t P[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u,
                                      0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr uint32_t INV = 0xefffffffu;  // -p^{-1} mod 2^32
    static constexpr uint32_t ONE[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u,
                                        0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    static constexpr uint32_t R2[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u,
                                       0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
};

struct FqParams {
    static constexpr uint32_t P[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u,
                                      0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    static constexpr uint32_t INV = 0xe4866389u;
    static constexpr uint32_t ONE[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u,
                                        0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
    static constexpr uint32_t R2[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u,
                                       0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
};

template <class PRM>
struct alignas(16) Fe {
    uint32_t v[8];

    __host__ __device__ __forceinline__ static Fe zero() {
        Fe r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.v[i] = 0;
        return r;
    }
    __host__ __device__ __forceinline__ static Fe one() {
        Fe r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.v[i] = PRM::ONE[i];
        return r;
    }
    __host__ __device__ __forceinline__ static Fe r2() {
        Fe r;
#pragma unroll
        for (int i = 0; i < 8; i++) r.v[i] = PRM::R2[i];
        return r;
    }
    __host__ __device__ __forceinline__ bool is_zero() const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) o |= v[i];
        return o == 0;
    }
    __host__ __device__ __forceinline__ bool operator==(const Fe& b) const {
        uint32_t o = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) o |= v[i] ^ b.v[i];
        return o == 0;
    }
    __host__ __device__ __forceinline__ bool operator!=(const Fe& b) const { return !(*this == b); }
};

// r = a - p if a >= p (a < 2p assumed)
template <class PRM>
__host__ __device__ __forceinline__ void reduce_once(Fe<PRM>& a) {
    uint32_t t[8];
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (int64_t)a.v[i] - (int64_t)PRM::P[i];
        t[i] = (uint32_t)c;
        c >>= 32;
    }
    if (c == 0) {  // no borrow: a >= p
#pragma unroll
        for (int i = 0; i < 8; i++) a.v[i] = t[i];
    }
}

// Portable add / sub (host, and reference for the device forms).
template <class PRM>
__host__ __device__ __forceinline__ Fe<PRM> fe_add_portable(const Fe<PRM>& a, const Fe<PRM>& b) {
    Fe<PRM> r;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (uint64_t)a.v[i] + b.v[i];
        r.v[i] = (uint32_t)c;
        c >>= 32;
    }
    reduce_once(r);  // p < 2^254: no carry out of 256 bits
    return r;
}

template <class PRM>
__host__ __device__ __forceinline__ Fe<PRM> fe_sub_portable(const Fe<PRM>& a, const Fe<PRM>& b) {
    Fe<PRM> r;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (int64_t)a.v[i] - (int64_t)b.v[i];
        r.v[i] = (uint32_t)c;
        c >>= 32;
    }
    if (c != 0) {  // borrow: add p back
        uint64_t d = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            d += (uint64_t)r.v[i] + PRM::P[i];
            r.v[i] = (uint32_t)d;
            d >>= 32;
        }
    }
    return r;
}

#if defined(__HIP_DEVICE_COMPILE__)
// gfx950 forms: straight carry chains (hipcc turns the portable 64-bit-accumulator code into
// ~90 instructions per addition, half of them moves).  add: r = a + b - p, then + (p & mask) if
// that borrowed; sub: r = a - b, then + (p & mask) if that borrowed.  r holds a on entry and the
// result on exit.  The modulus limbs are VGPR operands: a literal or SGPR together with the
// VCC carry-in would need two constant-bus reads, which a gfx9-family VOP2 does not have.
template <class PRM>
__device__ __forceinline__ void zk_add_asm(uint32_t (&r)[8], const uint32_t (&b)[8]) {
    uint32_t m, t;
    asm("v_add_co_u32 %0, vcc, %0, %10\n\tv_addc_co_u32 %1, vcc, %1, %11, vcc\n\tv_addc_co_u32 %2, vcc, %2, %12, vcc\n\tv_addc_co_u32 %3, vcc, %3, %13, vcc\n\tv_addc_co_u32 %4, vcc, %4, %14, vcc\n\tv_addc_co_u32 %5, vcc, %5, %15, vcc\n\tv_addc_co_u32 %6, vcc, %6, %16, vcc\n\tv_addc_co_u32 %7, vcc, %7, %17, vcc\n\tv_sub_co_u32 %0, vcc, %0, %18\n\tv_subb_co_u32 %1, vcc, %1, %19, vcc\n\tv_subb_co_u32 %2, vcc, %2, %20, vcc\n\tv_subb_co_u32 %3, vcc, %3, %21, vcc\n\tv_subb_co_u32 %4, vcc, %4, %22, vcc\n\tv_subb_co_u32 %5, vcc, %5, %23, vcc\n\tv_subb_co_u32 %6, vcc, %6, %24, vcc\n\tv_subb_co_u32 %7, vcc, %7, %25, vcc\n\tv_cndmask_b32_e64 %8, 0, -1, vcc\n\tv_and_b32 %9, %18, %8\n\tv_add_co_u32 %0, vcc, %0, %9\n\tv_and_b32 %9, %19, %8\n\tv_addc_co_u32 %1, vcc, %1, %9, vcc\n\tv_and_b32 %9, %20, %8\n\tv_addc_co_u32 %2, vcc, %2, %9, vcc\n\tv_and_b32 %9, %21, %8\n\tv_addc_co_u32 %3, vcc, %3, %9, vcc\n\tv_and_b32 %9, %22, %8\n\tv_addc_co_u32 %4, vcc, %4, %9, vcc\n\tv_and_b32 %9, %23, %8\n\tv_addc_co_u32 %5, vcc, %5, %9, vcc\n\tv_and_b32 %9, %24, %8\n\tv_addc_co_u32 %6, vcc, %6, %9, vcc\n\tv_and_b32 %9, %25, %8\n\tv_addc_co_u32 %7, vcc, %7, %9, vcc"
        : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "=&v"(m), "=&v"(t)
        : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "v"(PRM::P[0]), "v"(PRM::P[1]),
          "v"(PRM::P[2]), "v"(PRM::P[3]), "v"(PRM::P[4]), "v"(PRM::P[5]), "v"(PRM::P[6]), "v"(PRM::P[7])
        : "vcc");
}
template <class PRM>
__device__ __forceinline__ void zk_sub_asm(uint32_t (&r)[8], const uint32_t (&b)[8]) {
    uint32_t m, t;
    asm("v_sub_co_u32 %0, vcc, %0, %10\n\tv_subb_co_u32 %1, vcc, %1, %11, vcc\n\tv_subb_co_u32 %2, vcc, %2, %12, vcc\n\tv_subb_co_u32 %3, vcc, %3, %13, vcc\n\tv_subb_co_u32 %4, vcc, %4, %14, vcc\n\tv_subb_co_u32 %5, vcc, %5, %15, vcc\n\tv_subb_co_u32 %6, vcc, %6, %16, vcc\n\tv_subb_co_u32 %7, vcc, %7, %17, vcc\n\tv_cndmask_b32_e64 %8, 0, -1, vcc\n\tv_and_b32 %9, %18, %8\n\tv_add_co_u32 %0, vcc, %0, %9\n\tv_and_b32 %9, %19, %8\n\tv_addc_co_u32 %1, vcc, %1, %9, vcc\n\tv_and_b32 %9, %20, %8\n\tv_addc_co_u32 %2, vcc, %2, %9, vcc\n\tv_and_b32 %9, %21, %8\n\tv_addc_co_u32 %3, vcc, %3, %9, vcc\n\tv_and_b32 %9, %22, %8\n\tv_addc_co_u32 %4, vcc, %4, %9, vcc\n\tv_and_b32 %9, %23, %8\n\tv_addc_co_u32 %5, vcc, %5, %9, vcc\n\tv_and_b32 %9, %24, %8\n\tv_addc_co_u32 %6, vcc, %6, %9, vcc\n\tv_and_b32 %9, %25, %8\n\tv_addc_co_u32 %7, vcc, %7, %9, vcc"
        : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "=&v"(m), "=&v"(t)
        : "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]), "v"(PRM::P[0]), "v"(PRM::P[1]),
          "v"(PRM::P[2]), "v"(PRM::P[3]), "v"(PRM::P[4]), "v"(PRM::P[5]), "v"(PRM::P[6]), "v"(PRM::P[7])
        : "vcc");
}
#endif

template <class PRM>
__host__ __device__ __forceinline__ Fe<PRM> fe_add(const Fe<PRM>& a, const Fe<PRM>& b) {
#if defined(__HIP_DEVICE_COMPILE__)
    Fe<PRM> r = a;
    zk_add_asm<PRM>(r.v, b.v);
    return r;
#else
    return fe_add_portable(a, b);
#endif
}

template <class PRM>
__host__ __device__ __forceinline__ Fe<PRM> fe_sub(const Fe<PRM>& a, const Fe<PRM>& b) {
#if defined(__HIP_DEVICE_COMPILE__)
    Fe<PRM> r = a;
    zk_sub_asm<PRM>(r.v, b.v);
    return r;
#else
    return fe_sub_portable(a, b);
#endif
}

template <class PRM>
__host__ __device__ __forceinline__ Fe<PRM> fe_neg(const Fe<PRM>& a) {
    if (a.is_zero()) return a;
    Fe<PRM> r;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        c += (int64_t)PRM::P[i] - (int64_t)a.v[i];
        r.v[i] = (uint32_t)c;
        c >>= 32;
    }
    return r;
}

template <class PRM>
__host__ __device__ __forceinline__ Fe<PRM> fe_dbl(const Fe<PRM>& a) { return fe_add(a, a); }

// CIOS Montgomery product, 8 x 32-bit limbs (portable form: host code, and the reference
// against which the device form below is tested).  p < 2^254 so the running value stays
// below 2p * 2^32 and the 10th limb of the textbook algorithm is always 0.
template <class PRM>
__host__ __device__ __forceinline__ Fe<PRM> fe_mul_portable(const Fe<PRM>& a, const Fe<PRM>& b) {
    uint32_t t[9];
#pragma unroll
    for (int i = 0; i < 9; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t c = 0;
        const uint32_t bi = b.v[i];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            c += (uint64_t)a.v[j] * bi + t[j];
            t[j] = (uint32_t)c;
            c >>= 32;
        }
        const uint32_t t8 = t[8] + (uint32_t)c;
        const uint32_t m = t[0] * PRM::INV;
        c = (uint64_t)m * PRM::P[0] + t[0];
        c >>= 32;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            c += (uint64_t)m * PRM::P[j] + t[j];
            t[j - 1] = (uint32_t)c;
            c >>= 32;
        }
        c += t8;
        t[7] = (uint32_t)c;
        t[8] = (uint32_t)(c >> 32);
    }
    Fe<PRM> r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = t[i];
    reduce_once(r);
    return r;
}

#if !defined(__HIP_DEVICE_COMPILE__)
// Host (x86-64) form: the same CIOS on 4 x 64-bit limbs with 128-bit products — the transcript-side
// arithmetic (window Horner, to_affine, Lagrange interpolation) sits between GPU phases of a proof.
template <class PRM>
inline Fe<PRM> fe_mul_host64(const Fe<PRM>& a, const Fe<PRM>& b) {
    typedef unsigned __int128 u128;
    uint64_t A[4], B[4], P[4], t[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        A[i] = (uint64_t)a.v[2 * i] | ((uint64_t)a.v[2 * i + 1] << 32);
        B[i] = (uint64_t)b.v[2 * i] | ((uint64_t)b.v[2 * i + 1] << 32);
        P[i] = (uint64_t)PRM::P[2 * i] | ((uint64_t)PRM::P[2 * i + 1] << 32);
    }
    // -p^-1 mod 2^64 from the 32-bit constant: one Newton step
    const uint64_t inv32 = PRM::INV;
    const uint64_t inv64 = inv32 * (2 + P[0] * inv32);
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)A[j] * B[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        const uint64_t t4 = t[4] + (uint64_t)c;
        const uint64_t m = t[0] * inv64;
        c = ((u128)m * P[0] + t[0]) >> 64;
        for (int j = 1; j < 4; j++) {
            c += (u128)m * P[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t4;
        t[3] = (uint64_t)c;
        t[4] = (uint64_t)(c >> 64);
    }
    Fe<PRM> r;
    for (int i = 0; i < 4; i++) {
        r.v[2 * i] = (uint32_t)t[i];
        r.v[2 * i + 1] = (uint32_t)(t[i] >> 32);
    }
    reduce_once(r);
    return r;
}
#endif

#if defined(__HIP_DEVICE_COMPILE__)
// gfx950 form: product-scanning (FIPS) Montgomery.  Column k of a*b + m*p is summed in a
// 96-bit accumulator (lo: 64-bit VGPR pair, hi: carry count).  One product = v_mad_u64_u32
// (32x32 + 64 -> 64, carry-out in VCC) + v_addc_co_u32 — hipcc never uses the mad's carry-out
// by itself (it emits mad + 64-bit add + compare + select, ~4.5 instructions per product, half of
// them register moves), hence the two-instruction asm.  The first product of a column cannot
// overflow (the shifted-in accumulator is < 2^37) and needs no carry instruction.
// 128 mads + 112 carries + 8 mul_lo per product instead of ~580 instructions.
// zk_macN: N products accumulated in ONE asm statement (hipcc pads every asm boundary with an
// s_nop, so the products of a column are chained in groups of up to four).
#define ZK_MAC_BODY1 "v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
__device__ __forceinline__ void zk_mac1(uint64_t& lo, uint32_t& hi, uint32_t x0, uint32_t y0) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(hi) : "v"(x0), "v"(y0) : "vcc");
}
__device__ __forceinline__ void zk_mac2(uint64_t& lo, uint32_t& hi, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(hi) : "v"(x0), "v"(y0), "v"(x1), "v"(y1) : "vcc");
}
__device__ __forceinline__ void zk_mac4(uint64_t& lo, uint32_t& hi, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
                                        uint32_t x2, uint32_t y2, uint32_t x3, uint32_t y3) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %6, %7, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "+v"(lo), "+v"(hi) : "v"(x0), "v"(y0), "v"(x1), "v"(y1), "v"(x2), "v"(y2), "v"(x3), "v"(y3) : "vcc");
}
// same with the second factor a uniform constant (modulus limb) in an SGPR
__device__ __forceinline__ void zk_mac1k(uint64_t& lo, uint32_t& hi, uint32_t x0, uint32_t k0) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(hi) : "v"(x0), "s"(k0) : "vcc");
}
__device__ __forceinline__ void zk_mac2k(uint64_t& lo, uint32_t& hi, uint32_t x0, uint32_t k0, uint32_t x1, uint32_t k1) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc" : "+v"(lo), "+v"(hi) : "v"(x0), "s"(k0), "v"(x1), "s"(k1) : "vcc");
}
__device__ __forceinline__ void zk_mac4k(uint64_t& lo, uint32_t& hi, uint32_t x0, uint32_t k0, uint32_t x1, uint32_t k1,
                                         uint32_t x2, uint32_t k2, uint32_t x3, uint32_t k3) {
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %4, %5, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %6, %7, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc\n\tv_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
        : "+v"(lo), "+v"(hi) : "v"(x0), "s"(k0), "v"(x1), "s"(k1), "v"(x2), "s"(k2), "v"(x3), "s"(k3) : "vcc");
}
// first product of a column: the shifted-in accumulator is < 2^37, no carry possible
__device__ __forceinline__ void zk_mac_nc(uint64_t& lo, uint32_t x, uint32_t y) {
    asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(lo) : "v"(x), "v"(y) : "vcc");
}

// sum_{i = I0}^{I1-1} a[i] * b[K - i]
template <int I0, int I1, int K, class PRM>
__device__ __forceinline__ void zk_col_ab(uint64_t& lo, uint32_t& hi, const Fe<PRM>& a, const Fe<PRM>& b) {
    if constexpr (I1 - I0 >= 4) {
        zk_mac4(lo, hi, a.v[I0], b.v[K - I0], a.v[I0 + 1], b.v[K - I0 - 1], a.v[I0 + 2], b.v[K - I0 - 2], a.v[I0 + 3], b.v[K - I0 - 3]);
        zk_col_ab<I0 + 4, I1, K>(lo, hi, a, b);
    } else if constexpr (I1 - I0 >= 2) {
        zk_mac2(lo, hi, a.v[I0], b.v[K - I0], a.v[I0 + 1], b.v[K - I0 - 1]);
        zk_col_ab<I0 + 2, I1, K>(lo, hi, a, b);
    } else if constexpr (I1 - I0 == 1) {
        zk_mac1(lo, hi, a.v[I0], b.v[K - I0]);
    }
}
// sum_{i = I0}^{I1-1} m[i] * P[K - i]
template <int I0, int I1, int K, class PRM>
__device__ __forceinline__ void zk_col_mp(uint64_t& lo, uint32_t& hi, const uint32_t (&m)[8]) {
    if constexpr (I1 - I0 >= 4) {
        zk_mac4k(lo, hi, m[I0], PRM::P[K - I0], m[I0 + 1], PRM::P[K - I0 - 1], m[I0 + 2], PRM::P[K - I0 - 2], m[I0 + 3], PRM::P[K - I0 - 3]);
        zk_col_mp<I0 + 4, I1, K, PRM>(lo, hi, m);
    } else if constexpr (I1 - I0 >= 2) {
        zk_mac2k(lo, hi, m[I0], PRM::P[K - I0], m[I0 + 1], PRM::P[K - I0 - 1]);
        zk_col_mp<I0 + 2, I1, K, PRM>(lo, hi, m);
    } else if constexpr (I1 - I0 == 1) {
        zk_mac1k(lo, hi, m[I0], PRM::P[K - I0]);
    }
}

template <int K, class PRM>
__device__ __forceinline__ void zk_fips_low(uint64_t& lo, uint32_t& hi, uint32_t (&m)[8], const Fe<PRM>& a, const Fe<PRM>& b) {
    zk_mac_nc(lo, a.v[0], b.v[K]);
    zk_col_ab<1, K + 1, K>(lo, hi, a, b);
    zk_col_mp<0, K, K, PRM>(lo, hi, m);
    m[K] = (uint32_t)lo * PRM::INV;
    zk_mac1k(lo, hi, m[K], PRM::P[0]);
    lo = (lo >> 32) | ((uint64_t)hi << 32);
    hi = 0;
}
template <int K, class PRM>
__device__ __forceinline__ void zk_fips_high(uint64_t& lo, uint32_t& hi, const uint32_t (&m)[8], Fe<PRM>& r, const Fe<PRM>& a,
                                             const Fe<PRM>& b) {
    zk_mac_nc(lo, a.v[K - 7], b.v[7]);
    zk_col_ab<K - 6, 8, K>(lo, hi, a, b);
    zk_col_mp<K - 7, 8, K, PRM>(lo, hi, m);
    r.v[K - 8] = (uint32_t)lo;
    lo = (lo >> 32) | ((uint64_t)hi << 32);
    hi = 0;
}

// r in [0, 2p) -> [0, p): subtract p, add it back under the borrow mask (25 instructions)
template <class PRM>
__device__ __forceinline__ void zk_reduce_once_asm(uint32_t (&r)[8]) {
    uint32_t m, t;
    asm("v_sub_co_u32 %0, vcc, %0, %10\n\tv_subb_co_u32 %1, vcc, %1, %11, vcc\n\tv_subb_co_u32 %2, vcc, %2, %12, vcc\n\tv_subb_co_u32 %3, vcc, %3, %13, vcc\n\tv_subb_co_u32 %4, vcc, %4, %14, vcc\n\tv_subb_co_u32 %5, vcc, %5, %15, vcc\n\tv_subb_co_u32 %6, vcc, %6, %16, vcc\n\tv_subb_co_u32 %7, vcc, %7, %17, vcc\n\tv_cndmask_b32_e64 %8, 0, -1, vcc\n\tv_and_b32 %9, %10, %8\n\tv_add_co_u32 %0, vcc, %0, %9\n\tv_and_b32 %9, %11, %8\n\tv_addc_co_u32 %1, vcc, %1, %9, vcc\n\tv_and_b32 %9, %12, %8\n\tv_addc_co_u32 %2, vcc, %2, %9, vcc\n\tv_and_b32 %9, %13, %8\n\tv_addc_co_u32 %3, vcc, %3, %9, vcc\n\tv_and_b32 %9, %14, %8\n\tv_addc_co_u32 %4, vcc, %4, %9, vcc\n\tv_and_b32 %9, %15, %8\n\tv_addc_co_u32 %5, vcc, %5, %9, vcc\n\tv_and_b32 %9, %16, %8\n\tv_addc_co_u32 %6, vcc, %6, %9, vcc\n\tv_and_b32 %9, %17, %8\n\tv_addc_co_u32 %7, vcc, %7, %9, vcc"
        : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "=&v"(m), "=&v"(t)
        : "v"(PRM::P[0]), "v"(PRM::P[1]), "v"(PRM::P[2]), "v"(PRM::P[3]), "v"(PRM::P[4]), "v"(PRM::P[5]), "v"(PRM::P[6]), "v"(PRM::P[7])
        : "vcc");
}

template <class PRM>
__device__ __forceinline__ Fe<PRM> fe_mul_gfx950(const Fe<PRM>& a, const Fe<PRM>& b) {
    uint32_t m[8];
    Fe<PRM> r;
    uint64_t lo = 0;
    uint32_t hi = 0;
    zk_fips_low<0>(lo, hi, m, a, b);
    zk_fips_low<1>(lo, hi, m, a, b);
    zk_fips_low<2>(lo, hi, m, a, b);
    zk_fips_low<3>(lo, hi, m, a, b);
    zk_fips_low<4>(lo, hi, m, a, b);
    zk_fips_low<5>(lo, hi, m, a, b);
    zk_fips_low<6>(lo, hi, m, a, b);
    zk_fips_low<7>(lo, hi, m, a, b);
    zk_fips_high<8>(lo, hi, m, r, a, b);
    zk_fips_high<9>(lo, hi, m, r, a, b);
    zk_fips_high<10>(lo, hi, m, r, a, b);
    zk_fips_high<11>(lo, hi, m, r, a, b);
    zk_fips_high<12>(lo, hi, m, r, a, b);
    zk_fips_high<13>(lo, hi, m, r, a, b);
    zk_fips_high<14>(lo, hi, m, r, a, b);
    r.v[7] = (uint32_t)lo;  // column 15 is empty; the total is < 2p < 2^255
    zk_reduce_once_asm<PRM>(r.v);
    return r;
}
#endif

template <class PRM>
__host__ __device__ __forceinline__ Fe<PRM> fe_mul(const Fe<PRM>& a, const Fe<PRM>& b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return fe_mul_gfx950(a, b);
#else
    return fe_mul_host64(a, b);
#endif
}

template <class PRM>
__host__ __device__ __forceinline__ Fe<PRM> fe_sqr(const Fe<PRM>& a) { return fe_mul(a, a); }

// Montgomery -> canonical integer (multiply by 1)
template <class PRM>
__host__ __device__ __forceinline__ Fe<PRM> fe_from_mont(const Fe<PRM>& a) {
    Fe<PRM> one = Fe<PRM>::zero();
    one.v[0] = 1;
    return fe_mul(a, one);
}

template <class PRM>
__host__ __device__ __forceinline__ Fe<PRM> fe_to_mont(const Fe<PRM>& a) { return fe_mul(a, Fe<PRM>::r2()); }

// a^e, e given as 8 LE 32-bit words (canonical integer)
template <class PRM>
__host__ __device__ inline Fe<PRM> fe_pow(const Fe<PRM>& a, const uint32_t e[8]) {
    Fe<PRM> acc = Fe<PRM>::one();
    bool started = false;
    for (int i = 255; i >= 0; i--) {
        if (started) acc = fe_sqr(acc);
        if ((e[i >> 5] >> (i & 31)) & 1) {
            acc = started ? fe_mul(acc, a) : a;
            started = true;
        }
    }
    return acc;
}

template <class PRM>
__host__ __device__ inline Fe<PRM> fe_inv(const Fe<PRM>& a) {
    uint32_t e[8];
#pragma unroll
    for (int i = 0; i < 8; i++) e[i] = PRM::P[i];
    e[0] -= 2;  // p - 2 (low limb of both moduli is > 2)
    return fe_pow(a, e);
}

using Fr = Fe<FrParams>;
using Fq = Fe<FqParams>;

// 32-byte global loads/stores as two 16-byte accesses
template <class PRM>
__device__ __forceinline__ Fe<PRM> fe_load(const Fe<PRM>* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint4 lo = q[0], hi = q[1];
    Fe<PRM> r;
    r.v[0] = lo.x; r.v[1] = lo.y; r.v[2] = lo.z; r.v[3] = lo.w;
    r.v[4] = hi.x; r.v[5] = hi.y; r.v[6] = hi.z; r.v[7] = hi.w;
    return r;
}

template <class PRM>
__device__ __forceinline__ void fe_store(Fe<PRM>* p, const Fe<PRM>& a) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
    q[1] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
}

}  // namespace zk
