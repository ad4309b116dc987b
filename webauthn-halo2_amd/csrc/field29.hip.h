// field29.hip.h — carry-free field arithmetic for the hottest loops: 9 limbs of 29 bits.
//
// CDNA4's only wide multiplier is v_mad_u64_u32 (32 x 32 + 64 -> 64) and it has no carry-IN, so
// with saturated 32-bit limbs every product needs a second, equally slow, carry instruction
// (field.hip.h).  With 29-bit limbs a whole column of the product-scanning Montgomery product —
// nine a_i*b_j and nine m_i*p_j terms, each < 2^61 — fits one 64-bit accumulator: no carries, plain
// C++ (hipcc emits one v_mad_u64_u32 per term), 258 instead of 318 instructions per product and no
// dependent carry chain: 157 G products/s against 121 G/s chip-wide, and 1.5x faster for a lone wave
// (tools/ubench_f29.hip).
//
// Representation ("internal form"): x is held as X = x * 2^261 mod p, as an integer in [0, k*p) for a
// small k that the caller tracks ("lazy"), in limbs that are at most a little above 29 bits:
//   mul29   limbs a_i * b_j < 2^60.6 (e.g. 2^30.6 x 2^30), value bounds k_a * k_b <= 168 (2^261 / p =
//           169.4): result limbs < 2^29, value < 2p
//   add29   limb-wise;  sub29<K, E>  a + C - b with C = K*p spread so that every limb of C is >= 2^E:
//           needs limbs b_i < 2^E, value b < K*p (and b's top limb <= (K-1)p's); result value < a + K*p
//   norm29  carry propagation back to 29-bit limbs (value unchanged)
// The memory image of the rest of the engine (standard Montgomery form x * 2^256, 8 x 32-bit words, the
// Rust layout) converts for free on the way in: to29_x32 reads the limbs of (v << 5), i.e. 32 * v, a
// valid internal form with k = 32.  On the way out one product with the standard "one" (2^256) divides
// by 32 again.
#pragma once
#include "field.hip.h"

namespace zk {

static constexpr uint32_t M29 = (1u << 29) - 1;

template <class PRM>
struct Lim29 {
    // limb i of K * p (K small): 29-bit digits of the 256-bit product
    static constexpr uint32_t kp_limb(uint32_t K, int i) {
        uint64_t w[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        uint64_t carry = 0;
        for (int j = 0; j < 8; j++) {
            const uint64_t t = (uint64_t)PRM::P[j] * K + carry;
            w[j] = t & 0xffffffffu;
            carry = t >> 32;
        }
        w[8] = carry;
        const int bit = 29 * i, q = bit >> 5, o = bit & 31;
        uint64_t two = w[q];
        if (q + 1 < 9) two |= w[q + 1] << 32;
        return (uint32_t)(two >> o) & M29;
    }
    static constexpr uint32_t INV = PRM::INV & M29;  // -p^-1 mod 2^29
    // limb i of the spread form of K * p: every limb but the top carries an extra 2^E, borrowed from the next
    static constexpr uint32_t spread(uint32_t K, int E, int i) {
        const uint32_t d = kp_limb(K, i);
        const uint32_t borrow = 1u << (E - 29);
        if (i == 0) return d + (1u << E);
        if (i < 8) return d + (1u << E) - borrow;
        return d - borrow;
    }
    static constexpr uint32_t P[9] = {kp_limb(1, 0), kp_limb(1, 1), kp_limb(1, 2), kp_limb(1, 3), kp_limb(1, 4),
                                      kp_limb(1, 5), kp_limb(1, 6), kp_limb(1, 7), kp_limb(1, 8)};
};
template <class PRM, uint32_t K, int E>
struct Spread29 {
    typedef Lim29<PRM> L;
    static constexpr uint32_t C[9] = {L::spread(K, E, 0), L::spread(K, E, 1), L::spread(K, E, 2), L::spread(K, E, 3), L::spread(K, E, 4),
                                      L::spread(K, E, 5), L::spread(K, E, 6), L::spread(K, E, 7), L::spread(K, E, 8)};
};

template <class PRM>
struct Fe29 {
    uint32_t l[9];
};

// plain limb split of a 256-bit integer (value unchanged)
template <class PRM>
__device__ __forceinline__ Fe29<PRM> to29(const Fe<PRM>& a) {
    Fe29<PRM> r;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const int bit = 29 * i, w = bit >> 5, o = bit & 31;
        uint64_t two = a.v[w];
        if (w + 1 < 8) two |= (uint64_t)a.v[w + 1] << 32;
        r.l[i] = (uint32_t)(two >> o) & M29;
    }
    return r;
}

// limbs of 32 * a (a < 2^254): standard Montgomery form -> internal form with k = 32, limbs < 2^29
template <class PRM>
__device__ __forceinline__ Fe29<PRM> to29_x32(const Fe<PRM>& a) {
    Fe29<PRM> r;
    r.l[0] = (a.v[0] << 5) & M29;
#pragma unroll
    for (int i = 1; i < 9; i++) {
        const int bit = 29 * i - 5, w = bit >> 5, o = bit & 31;
        uint64_t two = a.v[w];
        if (w + 1 < 8) two |= (uint64_t)a.v[w + 1] << 32;
        r.l[i] = (uint32_t)(two >> o) & M29;
    }
    return r;
}

// normalised limbs (< 2^29), value < 2^256 -> 8 x 32-bit words
template <class PRM>
__device__ __forceinline__ Fe<PRM> from29(const Fe29<PRM>& a) {
    Fe<PRM> r;
#pragma unroll
    for (int w = 0; w < 8; w++) {
        const int i = (32 * w) / 29, o = 32 * w - 29 * i;
        uint64_t v = (uint64_t)a.l[i] >> o;
        v |= (uint64_t)a.l[i + 1] << (29 - o);
        if (i + 2 < 9) v |= (uint64_t)a.l[i + 2] << (58 - o);
        r.v[w] = (uint32_t)v;
    }
    return r;
}

// acc += a * b.  hipcc starts every column of a product on a fresh accumulator (independent of the previous column's carry)
// and joins the two by a 64-bit addition: 16 half-rate instructions per product that a strictly serial column does not
// need, in exchange for multiply-adds that do not wait for each other.  ZK_MUL29_ASM = 1 (set per file, before this
// header) pins the source order with one asm statement per multiply-add: right for kernels that run four waves per SIMD
// (the bucket accumulation: +3.4 % proofs/s), neutral or worse elsewhere (quotient, NTT: 3 waves per SIMD).
#ifndef ZK_MUL29_ASM
#define ZK_MUL29_ASM 0
#endif
// The products take the choice as a template parameter SER (default: the file's ZK_MUL29_ASM): kernels that run ONE wave per
// SIMD — the reduction tails of the MSM — are bound by the latency of that serial chain, not by issue slots, and want the
// compiler's form (msm_wrowcol / msm_wbits: -22 % with SER = false in a file whose accumulation kernels need SER = true).
#if defined(__HIP_DEVICE_COMPILE__)
template <bool SER>
__device__ __forceinline__ void mad29(uint64_t& acc, uint32_t a, uint32_t b) {
    if constexpr (SER) asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b) : "vcc");
    else acc += (uint64_t)a * b;
}
template <bool SER>
__device__ __forceinline__ void mad29c(uint64_t& acc, uint32_t a, uint32_t c) {
    if constexpr (SER) asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(a), "s"(c) : "vcc");
    else acc += (uint64_t)a * c;
}
#else
template <bool SER>
__device__ __forceinline__ void mad29(uint64_t& acc, uint32_t a, uint32_t b) { acc += (uint64_t)a * b; }
template <bool SER>
__device__ __forceinline__ void mad29c(uint64_t& acc, uint32_t a, uint32_t c) { acc += (uint64_t)a * c; }
#endif
static constexpr bool MUL29_SER = ZK_MUL29_ASM != 0;
// ZK_MUL29_ASM = 2: the serial form with a whole column's multiply-adds in ONE asm statement.  hipcc's hazard recogniser
// assumes that any inline asm may have the gfx940 "dst_sel forwarding" hazard and puts an s_nop behind every statement whose
// result the next instruction reads: 1 358 s_nop per mixed addition with one statement per multiply-add (hipcc -S of
// msm_wacc_fast_kernel), about 300 with one per column part.  v_mad_u64_u32 has no such hazard.
static constexpr bool MUL29_BLOCK = ZK_MUL29_ASM == 2;
#if defined(__HIP_DEVICE_COMPILE__)
// acc += sum_{i < cnt} x[i] * y[i], cnt <= 9 a constant after unrolling (the switch folds)
__device__ __forceinline__ void madcol_v(uint64_t& acc, int cnt, const uint32_t (&x)[9], const uint32_t (&y)[9]) {
    switch (cnt) {
    case 1: asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t" : "+v"(acc) : "v"(x[0]), "v"(y[0]) : "vcc"); break;
    case 2: asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t" "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t" : "+v"(acc) : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]) : "vcc"); break;
    case 3: asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t" "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t" "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t" : "+v"(acc) : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]) : "vcc"); break;
    case 4: asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t" "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t" "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t" "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t" : "+v"(acc) : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]) : "vcc"); break;
    case 5: asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t" "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t" "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t" "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t" "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\t" : "+v"(acc) : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]) : "vcc"); break;
    case 6: asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t" "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t" "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t" "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t" "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\t" "v_mad_u64_u32 %0, vcc, %11, %12, %0\n\t" : "+v"(acc) : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]) : "vcc"); break;
    case 7: asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t" "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t" "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t" "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t" "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\t" "v_mad_u64_u32 %0, vcc, %11, %12, %0\n\t" "v_mad_u64_u32 %0, vcc, %13, %14, %0\n\t" : "+v"(acc) : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]) : "vcc"); break;
    case 8: asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t" "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t" "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t" "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t" "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\t" "v_mad_u64_u32 %0, vcc, %11, %12, %0\n\t" "v_mad_u64_u32 %0, vcc, %13, %14, %0\n\t" "v_mad_u64_u32 %0, vcc, %15, %16, %0\n\t" : "+v"(acc) : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]), "v"(x[7]), "v"(y[7]) : "vcc"); break;
    case 9: asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t" "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t" "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t" "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t" "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\t" "v_mad_u64_u32 %0, vcc, %11, %12, %0\n\t" "v_mad_u64_u32 %0, vcc, %13, %14, %0\n\t" "v_mad_u64_u32 %0, vcc, %15, %16, %0\n\t" "v_mad_u64_u32 %0, vcc, %17, %18, %0\n\t" : "+v"(acc) : "v"(x[0]), "v"(y[0]), "v"(x[1]), "v"(y[1]), "v"(x[2]), "v"(y[2]), "v"(x[3]), "v"(y[3]), "v"(x[4]), "v"(y[4]), "v"(x[5]), "v"(y[5]), "v"(x[6]), "v"(y[6]), "v"(x[7]), "v"(y[7]), "v"(x[8]), "v"(y[8]) : "vcc"); break;
    default: break;
    }
}
// the same with y in scalar registers (compile-time constants: the limbs of p)
__device__ __forceinline__ void madcol_c(uint64_t& acc, int cnt, const uint32_t (&x)[9], const uint32_t (&y)[9]) {
    switch (cnt) {
    case 1: asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t" : "+v"(acc) : "v"(x[0]), "s"(y[0]) : "vcc"); break;
    case 2: asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t" "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t" : "+v"(acc) : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]) : "vcc"); break;
    case 3: asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t" "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t" "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t" : "+v"(acc) : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]) : "vcc"); break;
    case 4: asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t" "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t" "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t" "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t" : "+v"(acc) : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]), "v"(x[3]), "s"(y[3]) : "vcc"); break;
    case 5: asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t" "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t" "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t" "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t" "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\t" : "+v"(acc) : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]), "v"(x[3]), "s"(y[3]), "v"(x[4]), "s"(y[4]) : "vcc"); break;
    case 6: asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t" "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t" "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t" "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t" "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\t" "v_mad_u64_u32 %0, vcc, %11, %12, %0\n\t" : "+v"(acc) : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]), "v"(x[3]), "s"(y[3]), "v"(x[4]), "s"(y[4]), "v"(x[5]), "s"(y[5]) : "vcc"); break;
    case 7: asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t" "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t" "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t" "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t" "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\t" "v_mad_u64_u32 %0, vcc, %11, %12, %0\n\t" "v_mad_u64_u32 %0, vcc, %13, %14, %0\n\t" : "+v"(acc) : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]), "v"(x[3]), "s"(y[3]), "v"(x[4]), "s"(y[4]), "v"(x[5]), "s"(y[5]), "v"(x[6]), "s"(y[6]) : "vcc"); break;
    case 8: asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t" "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t" "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t" "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t" "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\t" "v_mad_u64_u32 %0, vcc, %11, %12, %0\n\t" "v_mad_u64_u32 %0, vcc, %13, %14, %0\n\t" "v_mad_u64_u32 %0, vcc, %15, %16, %0\n\t" : "+v"(acc) : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]), "v"(x[3]), "s"(y[3]), "v"(x[4]), "s"(y[4]), "v"(x[5]), "s"(y[5]), "v"(x[6]), "s"(y[6]), "v"(x[7]), "s"(y[7]) : "vcc"); break;
    case 9: asm("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\t" "v_mad_u64_u32 %0, vcc, %3, %4, %0\n\t" "v_mad_u64_u32 %0, vcc, %5, %6, %0\n\t" "v_mad_u64_u32 %0, vcc, %7, %8, %0\n\t" "v_mad_u64_u32 %0, vcc, %9, %10, %0\n\t" "v_mad_u64_u32 %0, vcc, %11, %12, %0\n\t" "v_mad_u64_u32 %0, vcc, %13, %14, %0\n\t" "v_mad_u64_u32 %0, vcc, %15, %16, %0\n\t" "v_mad_u64_u32 %0, vcc, %17, %18, %0\n\t" : "+v"(acc) : "v"(x[0]), "s"(y[0]), "v"(x[1]), "s"(y[1]), "v"(x[2]), "s"(y[2]), "v"(x[3]), "s"(y[3]), "v"(x[4]), "s"(y[4]), "v"(x[5]), "s"(y[5]), "v"(x[6]), "s"(y[6]), "v"(x[7]), "s"(y[7]), "v"(x[8]), "s"(y[8]) : "vcc"); break;
    default: break;
    }
}
#else
__device__ __forceinline__ void madcol_v(uint64_t& acc, int cnt, const uint32_t (&x)[9], const uint32_t (&y)[9]) {
    for (int i = 0; i < cnt; i++) acc += (uint64_t)x[i] * y[i];
}
__device__ __forceinline__ void madcol_c(uint64_t& acc, int cnt, const uint32_t (&x)[9], const uint32_t (&y)[9]) {
    for (int i = 0; i < cnt; i++) acc += (uint64_t)x[i] * y[i];
}
#endif
// acc = x * y: the first multiply-add of a product takes the inline constant 0 as its addend instead of a zeroed accumulator
// (one v_mov_b64 — issued at the multiplier's rate — less per product)
__device__ __forceinline__ void mad_first(uint64_t& acc, uint32_t x, uint32_t y) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(acc) : "v"(x), "v"(y) : "vcc");
#else
    acc = (uint64_t)x * y;
#endif
}
// the result limbs' masks as ONE run of plain VOP2 instructions (ZK_MUL29_MASKRUN, experiment: a plain instruction between two
// multiply-adds costs 3.96 cycles, in a run 2.5 — profiles/r6_ubench_isa.txt): r[i] = raw[i] & (2^29 - 1), i < 8
#ifndef ZK_MUL29_MASKRUN
#define ZK_MUL29_MASKRUN 0
#endif
__device__ __forceinline__ void mask_run8(uint32_t (&r)[9], const uint32_t (&raw)[8]) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_and_b32 %0, 0x1fffffff, %8\n\tv_and_b32 %1, 0x1fffffff, %9\n\tv_and_b32 %2, 0x1fffffff, %10\n\tv_and_b32 %3, 0x1fffffff, %11\n\t"
        "v_and_b32 %4, 0x1fffffff, %12\n\tv_and_b32 %5, 0x1fffffff, %13\n\tv_and_b32 %6, 0x1fffffff, %14\n\tv_and_b32 %7, 0x1fffffff, %15"
        : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
        : "v"(raw[0]), "v"(raw[1]), "v"(raw[2]), "v"(raw[3]), "v"(raw[4]), "v"(raw[5]), "v"(raw[6]), "v"(raw[7]));
#else
    for (int i = 0; i < 8; i++) r[i] = raw[i] & M29;
#endif
}
// column helpers: sum_{i = lo}^{hi} a_i b_{k - i} and sum_{i = lo}^{hi} m_i p_{k - i}, one statement each in the block form
template <class PRM, bool SER>
__device__ __forceinline__ void col_ab(uint64_t& acc, const uint32_t (&a)[9], const uint32_t (&b)[9], int k, int lo, int hi) {
    if constexpr (SER && MUL29_BLOCK) {
        uint32_t x[9], y[9];
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int j = lo + i <= hi ? lo + i : lo;
            x[i] = a[j];
            y[i] = b[k - j];
        }
        madcol_v(acc, hi - lo + 1, x, y);
    } else {
#pragma unroll
        for (int i = lo; i <= hi; i++) mad29<SER>(acc, a[i], b[k - i]);
    }
}
template <class PRM, bool SER>
__device__ __forceinline__ void col_mp(uint64_t& acc, const uint32_t (&m)[9], int k, int lo, int hi) {
    if constexpr (SER && MUL29_BLOCK) {
        uint32_t x[9], y[9];
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int j = lo + i <= hi ? lo + i : lo;
            x[i] = m[j];
            y[i] = Lim29<PRM>::P[k - j];
        }
        madcol_c(acc, hi - lo + 1, x, y);
    } else {
#pragma unroll
        for (int i = lo; i <= hi; i++) mad29c<SER>(acc, m[i], Lim29<PRM>::P[k - i]);
    }
}

// a * b * 2^-261 mod p, lazily: result limbs < 2^29, value < p * (1 + k_a k_b / 169.4)
template <class PRM, bool SER = MUL29_SER>
__device__ __forceinline__ Fe29<PRM> mul29(const Fe29<PRM>& a, const Fe29<PRM>& b) {
    uint32_t m[9];
    Fe29<PRM> r;
    uint64_t acc = 0;
    uint32_t raw[8];
#pragma unroll
    for (int k = 0; k < 9; k++) {
        if (SER && MUL29_BLOCK && k == 0) mad_first(acc, a.l[0], b.l[0]);
        else col_ab<PRM, SER>(acc, a.l, b.l, k, 0, k);
        if (k) col_mp<PRM, SER>(acc, m, k, 0, k - 1);
        m[k] = ((uint32_t)acc * Lim29<PRM>::INV) & M29;
        mad29c<SER>(acc, m[k], Lim29<PRM>::P[0]);
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
        col_ab<PRM, SER>(acc, a.l, b.l, k, k - 8, 8);
        col_mp<PRM, SER>(acc, m, k, k - 8, 8);
        if (ZK_MUL29_MASKRUN && SER && MUL29_BLOCK) raw[k - 9] = (uint32_t)acc;
        else r.l[k - 9] = (uint32_t)acc & M29;
        acc >>= 29;
    }
    if (ZK_MUL29_MASKRUN && SER && MUL29_BLOCK) mask_run8(r.l, raw);
    r.l[8] = (uint32_t)acc;
    return r;
}

// (a * b + c * d) * 2^-261 mod p with ONE Montgomery reduction: both products are summed column by column before the
// m * p terms (243 instead of 324 multiply-adds).  All four operands need limbs <= ~2^30 so that a column
// (9 + 9 products + 9 reduction terms) stays below 2^64: a_i b_j + c_i d_j < 2^60.6 each side is too much — callers pass
// normalised (29-bit) a, c and b, d with limbs < 2^30.7.  Value bounds: k_a k_b + k_c k_d <= 168.
template <class PRM, bool SER = MUL29_SER>
__device__ __forceinline__ Fe29<PRM> mul2add29(const Fe29<PRM>& a, const Fe29<PRM>& b, const Fe29<PRM>& c, const Fe29<PRM>& d) {
    uint32_t m[9];
    Fe29<PRM> r;
    uint64_t acc = 0;
    uint32_t raw[8];
#pragma unroll
    for (int k = 0; k < 9; k++) {
        if (SER && MUL29_BLOCK && k == 0) mad_first(acc, a.l[0], b.l[0]);
        else col_ab<PRM, SER>(acc, a.l, b.l, k, 0, k);
        col_ab<PRM, SER>(acc, c.l, d.l, k, 0, k);
        if (k) col_mp<PRM, SER>(acc, m, k, 0, k - 1);
        m[k] = ((uint32_t)acc * Lim29<PRM>::INV) & M29;
        mad29c<SER>(acc, m[k], Lim29<PRM>::P[0]);
        acc >>= 29;
    }
#pragma unroll
    for (int k = 9; k < 17; k++) {
        col_ab<PRM, SER>(acc, a.l, b.l, k, k - 8, 8);
        col_ab<PRM, SER>(acc, c.l, d.l, k, k - 8, 8);
        col_mp<PRM, SER>(acc, m, k, k - 8, 8);
        if (ZK_MUL29_MASKRUN && SER && MUL29_BLOCK) raw[k - 9] = (uint32_t)acc;
        else r.l[k - 9] = (uint32_t)acc & M29;
        acc >>= 29;
    }
    if (ZK_MUL29_MASKRUN && SER && MUL29_BLOCK) mask_run8(r.l, raw);
    r.l[8] = (uint32_t)acc;
    return r;
}

// a * a * 2^-261 mod p: the 36 cross products are taken once against the doubled operand (45 instead of 81 products of a * a)
template <class PRM, bool SER = MUL29_SER>
__device__ __forceinline__ Fe29<PRM> sqr29(const Fe29<PRM>& a) {
    uint32_t m[9], a2[9], raw[8];
#pragma unroll
    for (int i = 0; i < 9; i++) a2[i] = a.l[i] << 1;
    Fe29<PRM> r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 17; k++) {
        // sum_{i + j = k, i < j} (2 a_i) a_j + [k even] a_{k/2}^2
        if constexpr (SER && MUL29_BLOCK) {
            uint32_t x[9], y[9];
            int cnt = 0;
#pragma unroll
            for (int i = (k > 8 ? k - 8 : 0); 2 * i < k; i++) {
                x[cnt] = a2[i];
                y[cnt] = a.l[k - i];
                cnt++;
            }
            if ((k & 1) == 0) {
                x[cnt] = a.l[k / 2];
                y[cnt] = a.l[k / 2];
                cnt++;
            }
            if (k == 0) mad_first(acc, a.l[0], a.l[0]);
            else madcol_v(acc, cnt, x, y);
        } else {
#pragma unroll
            for (int i = (k > 8 ? k - 8 : 0); 2 * i < k; i++) mad29<SER>(acc, a2[i], a.l[k - i]);
            if ((k & 1) == 0) mad29<SER>(acc, a.l[k / 2], a.l[k / 2]);
        }
        if (k < 9) {
            if (k) col_mp<PRM, SER>(acc, m, k, 0, k - 1);
            m[k] = ((uint32_t)acc * Lim29<PRM>::INV) & M29;
            mad29c<SER>(acc, m[k], Lim29<PRM>::P[0]);
        } else {
            col_mp<PRM, SER>(acc, m, k, k - 8, 8);
            if (ZK_MUL29_MASKRUN && SER && MUL29_BLOCK) raw[k - 9] = (uint32_t)acc;
            else r.l[k - 9] = (uint32_t)acc & M29;
        }
        acc >>= 29;
    }
    if (ZK_MUL29_MASKRUN && SER && MUL29_BLOCK) mask_run8(r.l, raw);
    r.l[8] = (uint32_t)acc;
    return r;
}

// sum_{j < K} a[j] * b[j] * 2^-261 mod p with ONE Montgomery reduction (81 (K + 1) instead of 162 K multiply-adds).
// All operands normalised (limbs < 2^29): a column is at most 9 K + 9 products < 2^58, so K <= 5.  Value bounds:
// sum_j k_a[j] k_b[j] <= 168 for a result < 2p.
template <int K, class PRM, bool SER = MUL29_SER>
__device__ __forceinline__ Fe29<PRM> mulKadd29(const Fe29<PRM> (&a)[K], const Fe29<PRM> (&b)[K]) {
    static_assert(K >= 1 && K <= 5, "a column of 9 K + 9 products of 58 bits must fit 64 bits");
    uint32_t m[9];
    Fe29<PRM> r;
    uint64_t acc = 0;
#pragma unroll
    for (int k = 0; k < 17; k++) {
#pragma unroll
        for (int j = 0; j < K; j++) {
#pragma unroll
            for (int i = (k > 8 ? k - 8 : 0); i <= (k < 9 ? k : 8); i++) acc += (uint64_t)a[j].l[i] * b[j].l[k - i];
        }
        if (k < 9) {
#pragma unroll
            for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * Lim29<PRM>::P[k - i];
            m[k] = ((uint32_t)acc * Lim29<PRM>::INV) & M29;
            acc += (uint64_t)m[k] * Lim29<PRM>::P[0];
        } else {
#pragma unroll
            for (int i = k - 8; i < 9; i++) acc += (uint64_t)m[i] * Lim29<PRM>::P[k - i];
            r.l[k - 9] = (uint32_t)acc & M29;
        }
        acc >>= 29;
    }
    r.l[8] = (uint32_t)acc;
    return r;
}

template <class PRM>
__device__ __forceinline__ Fe29<PRM> add29(const Fe29<PRM>& a, const Fe29<PRM>& b) {
    Fe29<PRM> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
}

// a - b + K*p, limb-wise (see the header for the conditions on b)
template <uint32_t K, int E, class PRM>
__device__ __forceinline__ Fe29<PRM> sub29(const Fe29<PRM>& a, const Fe29<PRM>& b) {
    Fe29<PRM> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + (Spread29<PRM, K, E>::C[i] - b.l[i]);
    return r;
}

template <class PRM>
__device__ __forceinline__ Fe29<PRM> norm29(const Fe29<PRM>& a) {
    Fe29<PRM> r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint32_t t = a.l[i] + c;
        r.l[i] = t & M29;
        c = t >> 29;
    }
    r.l[8] = a.l[8] + c;
    return r;
}

// value < 2p, normalised limbs: is it 0 mod p?
template <class PRM>
__device__ __forceinline__ bool is_zero29(const Fe29<PRM>& a) {
    uint32_t any = 0, dif = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        any |= a.l[i];
        dif |= a.l[i] ^ Lim29<PRM>::P[i];
    }
    return any == 0 || dif == 0;
}

// standard Montgomery form (canonical) -> internal form with k < 2: one product by 2^266 mod p
template <class PRM>
struct Conv29 {
    // 2^e mod p as 8 words, by repeated doubling at compile time
    static constexpr void pow2_words(int e, uint32_t (&out)[8]) {
        uint32_t v[8] = {1, 0, 0, 0, 0, 0, 0, 0};
        for (int s = 0; s < e; s++) {
            uint32_t carry = 0;
            for (int j = 0; j < 8; j++) {
                const uint32_t n = (v[j] << 1) | carry;
                carry = v[j] >> 31;
                v[j] = n;
            }
            // p < 2^254 and v < p before doubling, so 2v < 2^255: no carry out; subtract p if >= p
            bool ge = true;
            for (int j = 7; j >= 0; j--) {
                if (v[j] != PRM::P[j]) {
                    ge = v[j] > PRM::P[j];
                    break;
                }
            }
            if (ge) {
                uint64_t borrow = 0;
                for (int j = 0; j < 8; j++) {
                    const uint64_t t = (uint64_t)v[j] - PRM::P[j] - borrow;
                    v[j] = (uint32_t)t;
                    borrow = (t >> 63) & 1;
                }
            }
        }
        for (int j = 0; j < 8; j++) out[j] = v[j];
    }
    static constexpr uint32_t pow2_limb(int e, int i) {
        uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        pow2_words(e, w);
        const int bit = 29 * i, q = bit >> 5, o = bit & 31;
        uint64_t two = w[q];
        if (q + 1 < 8) two |= (uint64_t)w[q + 1] << 32;
        return (uint32_t)(two >> o) & M29;
    }
};

template <class PRM, int E>
struct Pow2_29 {
    typedef Conv29<PRM> C;
    static constexpr uint32_t L[9] = {C::pow2_limb(E, 0), C::pow2_limb(E, 1), C::pow2_limb(E, 2), C::pow2_limb(E, 3), C::pow2_limb(E, 4),
                                      C::pow2_limb(E, 5), C::pow2_limb(E, 6), C::pow2_limb(E, 7), C::pow2_limb(E, 8)};
};
template <int E, class PRM>
__device__ __forceinline__ Fe29<PRM> const_pow2_29() {
    Fe29<PRM> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = Pow2_29<PRM, E>::L[i];
    return r;
}

// x * 2^256 (canonical words) -> x * 2^261, k < 2:  (v) * 2^266 * 2^-261 = 32 v
template <class PRM>
__device__ __forceinline__ Fe29<PRM> std_to_internal(const Fe<PRM>& a) {
    return mul29(to29(a), const_pow2_29<266, PRM>());
}
// x * 2^261 (k_a <= 168) -> x * 2^256 canonical words:  a * 2^256 * 2^-261 = a / 32
template <class PRM>
__device__ __forceinline__ Fe<PRM> internal_to_std(const Fe29<PRM>& a) {
    Fe<PRM> r = from29(mul29(a, const_pow2_29<256, PRM>()));
    reduce_once(r);
    return r;
}

}  // namespace zk
