// ec29.hip.h — XYZZ accumulation on the carry-free 29-bit-limb field (field29.hip.h).
//
// Coordinates are held in internal form (x * 2^261 mod p, lazy).  Invariants between additions:
//   X: limbs < 2^29, value < 9p      Y: limbs < 2^29, value < 5p      ZZ, ZZZ: product outputs (< 2p)
// The bounds in the comments are (value as a multiple of p ; limb bits).
#pragma once
#include "ec.hip.h"
#include "field29.hip.h"

namespace zk {

typedef Fe29<FqParams> Fq29;

struct G1X29 {
    Fq29 x, y, zz, zzz;
    bool inf;
};

__device__ __forceinline__ G1X g1x29_to_std(const G1X29& a) {
    if (a.inf) return G1X::identity();
    G1X r;
    r.x = internal_to_std(a.x);
    r.y = internal_to_std(a.y);
    r.zz = internal_to_std(a.zz);
    r.zzz = internal_to_std(a.zzz);
    return r;
}

__device__ __forceinline__ G1X29 g1x29_from_std(const G1X& a) {
    G1X29 r;
    r.inf = a.is_identity();
    r.x = std_to_internal(a.x);
    r.y = std_to_internal(a.y);
    r.zz = std_to_internal(a.zz);
    r.zzz = std_to_internal(a.zzz);
    return r;
}

// acc += (x, y): an affine point in the standard memory form (canonical words), not the identity.
// madd-2008-s.  Returns false when the addition is one of the exceptional cases (same x: a doubling
// or a cancellation), which the caller redoes on the general path; acc is then unchanged.  CHECK = false skips the test
// (and always returns true): the caller detects an exceptional step by ZZ = 0 at the end and redoes the whole run.
// INTERNAL: (x, y) are already in the internal form (x * 2^261 mod p, canonical words — the wide path's window tables,
// msm.hip): their plain limbs are the operand (bound 1 instead of 32) and starting a run costs no product, so that a lane
// can restart its running sum in the middle of its entries (a new bucket) at the price of a few moves.
template <bool CHECK = true, bool INTERNAL = false>
__device__ __forceinline__ bool g1x29_add_affine(G1X29& acc, const Fq& x, const Fq& y) {
    if (acc.inf) {
        acc.x = INTERNAL ? to29(x) : std_to_internal(x);  // (2 ; 29)
        acc.y = INTERNAL ? to29(y) : std_to_internal(y);
        acc.zz = const_pow2_29<261, FqParams>();  // 1 in internal form
        acc.zzz = acc.zz;
        acc.inf = false;
        return true;
    }
    const Fq29 x2 = INTERNAL ? to29(x) : to29_x32(x), y2 = INTERNAL ? to29(y) : to29_x32(y);  // (32 ; 29), internal: (1 ; 29)
    const Fq29 u2 = mul29(x2, acc.zz);                            // 32 * 2 = 64 <= 168
    const Fq29 s2 = mul29(y2, acc.zzz);
    const Fq29 p = norm29(sub29<10, 29>(u2, acc.x));              // (12 ; 29)   X < 9p
    const Fq29 r = norm29(sub29<6, 29>(s2, acc.y));               // (8 ; 29)    Y < 5p
#ifndef ZK_EC29_SQR
#define ZK_EC29_SQR 1
#endif
#ifndef ZK_EC29_FUSE
#define ZK_EC29_FUSE 1
#endif
#if ZK_EC29_SQR
    const Fq29 pp = sqr29(p);                                     // 144
#else
    const Fq29 pp = mul29(p, p);                                  // 144
#endif
    // p prime: P = 0 (mod p) <=> P^2 = 0.  Without CHECK the caller looks at ZZ afterwards: it is a product of the P^2's, zero
    // from the first exceptional step on
    if (CHECK && is_zero29(pp)) return false;
    const Fq29 ppp = mul29(p, pp);                                // 24
    const Fq29 q = mul29(acc.x, pp);                              // 18
#if ZK_EC29_SQR
    const Fq29 rr = sqr29(r);                                     // 64
#else
    const Fq29 rr = mul29(r, r);                                  // 64
#endif
    const Fq29 t = add29(ppp, add29(q, q));                       // (6 ; < 3 * 2^29)
    const Fq29 x3 = norm29(sub29<7, 31>(rr, t));                  // (9 ; 29)
    const Fq29 v = sub29<10, 29>(q, x3);                          // (12 ; 30.6)
#if ZK_EC29_FUSE
    // Y3 = R (Q - X3) - Y1 PPP as ONE reduction: R v + Y1 (3p - PPP); bounds 8 * 12 + 5 * 3 = 111 <= 168, columns
    // 9 (2^29 * 2^30.6) + 9 (2^29 * 2^30) + 9 * 2^58 < 2^64
    Fq29 zero;
#pragma unroll
    for (int i = 0; i < 9; i++) zero.l[i] = 0;
    acc.x = x3;
    acc.y = mul2add29(r, v, acc.y, sub29<3, 29>(zero, ppp));      // (2 ; 29)
#else
    const Fq29 t1 = mul29(r, v);                                  // 96 ; limbs 2^29 * 2^30.6
    const Fq29 t2 = mul29(acc.y, ppp);                            // 10
    acc.x = x3;
    acc.y = norm29(sub29<3, 29>(t1, t2));                         // (5 ; 29)
#endif
    acc.zz = mul29(acc.zz, pp);
    acc.zzz = mul29(acc.zzz, ppp);
    return true;
}

// ---- partial sums kept in internal form between the accumulation and the reduction tails (msm.hip) ----
// Memory image of a G1X29: 4 x 9 limbs (144 bytes), every coordinate with normalised limbs (< 2^29) and the bounds
// of the header; the identity is all-zero (a point that is not the identity has ZZ != 0, hence a non-zero limb).
struct alignas(16) G1X29S {
    uint32_t w[36];
};

__device__ __forceinline__ G1X29 g1x29_load(const G1X29S* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    uint32_t w[36];
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const uint4 v = q[i];
        w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
    }
    G1X29 r;
    uint32_t any = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        r.x.l[i] = w[i];
        r.y.l[i] = w[9 + i];
        r.zz.l[i] = w[18 + i];
        r.zzz.l[i] = w[27 + i];
        any |= w[18 + i];
    }
    r.inf = any == 0;
    return r;
}

__device__ __forceinline__ void g1x29_store(G1X29S* p, const G1X29& a) {
    uint4* q = reinterpret_cast<uint4*>(p);
    uint32_t w[36];
#pragma unroll
    for (int i = 0; i < 9; i++) {
        w[i] = a.inf ? 0u : a.x.l[i];
        w[9 + i] = a.inf ? 0u : a.y.l[i];
        w[18 + i] = a.inf ? 0u : a.zz.l[i];
        w[27 + i] = a.inf ? 0u : a.zzz.l[i];
    }
#pragma unroll
    for (int i = 0; i < 9; i++) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}

__device__ __forceinline__ G1X29 g1x29_identity() {
    G1X29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.x.l[i] = r.y.l[i] = r.zz.l[i] = r.zzz.l[i] = 0;
    r.inf = true;
    return r;
}

// out-of-line product for the rare paths; vector-typed arguments travel in registers (a struct of nine words would
// go through the stack, i.e. scratch memory)
typedef uint32_t Limbs9 __attribute__((ext_vector_type(9)));
typedef uint32_t Words8 __attribute__((ext_vector_type(8)));
__device__ __noinline__ Limbs9 mul29_vcall(Limbs9 a, Limbs9 b) {
    Fq29 x, y;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        x.l[i] = a[i];
        y.l[i] = b[i];
    }
    const Fq29 z = mul29(x, y);
    Limbs9 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r[i] = z.l[i];
    return r;
}
__device__ __forceinline__ Fq29 mul29_call(const Fq29& a, const Fq29& b) {
    Limbs9 x, y;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        x[i] = a.l[i];
        y[i] = b.l[i];
    }
    const Limbs9 z = mul29_vcall(x, y);
    Fq29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = z[i];
    return r;
}

__device__ __forceinline__ Fq29 zero29() {
    Fq29 z;
#pragma unroll
    for (int i = 0; i < 9; i++) z.l[i] = 0;
    return z;
}

// 2 * p (dbl-2008-s-1, a = 0) on the lazy field; rare (two equal partial sums): inlined, but its products are calls.
// In: X (9 ; 29), Y (5 ; 29), ZZ, ZZZ (2 ; 29).  Out: X (7 ; 29), Y (5 ; 29), ZZ, ZZZ (2 ; 29).
__device__ __forceinline__ void g1x29_dbl_rare(G1X29& p) {
    if (p.inf) return;
    const Fq29 u = add29(p.y, p.y);                                        // (10 ; 30)
    const Fq29 v = mul29_call(u, u);                                       // 100
    const Fq29 w = mul29_call(u, v);                                       // 20
    const Fq29 s = mul29_call(p.x, v);                                     // 18
    const Fq29 xx = mul29_call(p.x, p.x);                                  // 81
    const Fq29 m = norm29(add29(xx, add29(xx, xx)));                       // (6 ; 29)
    const Fq29 mm = mul29_call(m, m);                                      // 36
    const Fq29 x3 = norm29(sub29<5, 30>(mm, add29(s, s)));                 // (7 ; 29)
    const Fq29 t1 = mul29_call(m, sub29<8, 29>(s, x3));                    // 6 * 10 = 60
    const Fq29 t2 = mul29_call(w, p.y);                                    // 10
    p.zz = mul29_call(v, p.zz);
    p.zzz = mul29_call(w, p.zzz);
    p.x = x3;
    p.y = norm29(sub29<3, 29>(t1, t2));                                    // (5 ; 29)
}

// acc += b, both XYZZ partial sums in internal form (add-2008-s), all special cases handled; the products are inlined,
// so a kernel should have ONE call site (3 300 instructions).  Bounds in and out as in the header (X < 9p, Y < 5p).
// SER: the products' multiply-adds as one serial chain (field29.hip.h) — false in the kernels that run one wave per SIMD.
template <bool SER = MUL29_SER>
__device__ __forceinline__ void g1x29_add(G1X29& acc, const G1X29& b) {
    if (b.inf) return;
    if (acc.inf) {
        acc = b;
        return;
    }
    const Fq29 u1 = mul29<FqParams, SER>(acc.x, b.zz);                                    // 18
    const Fq29 u2 = mul29<FqParams, SER>(b.x, acc.zz);
    const Fq29 s1 = mul29<FqParams, SER>(acc.y, b.zzz);                                   // 10
    const Fq29 s2 = mul29<FqParams, SER>(b.y, acc.zzz);
    const Fq29 p = norm29(sub29<3, 29>(u2, u1));                           // (5 ; 29)
    const Fq29 r = norm29(sub29<3, 29>(s2, s1));                           // (5 ; 29)
    const Fq29 pp = sqr29<FqParams, SER>(p);                                              // 25
    const Fq29 rr = sqr29<FqParams, SER>(r);                                              // 25
    if (is_zero29(pp)) {                                                   // same x: doubling or cancellation
        if (is_zero29(rr)) g1x29_dbl_rare(acc);
        else acc = g1x29_identity();
        return;
    }
    const Fq29 ppp = mul29<FqParams, SER>(p, pp);                                         // 10
    const Fq29 q = mul29<FqParams, SER>(u1, pp);                                          // 4
    const Fq29 t = add29(ppp, add29(q, q));                                // (6 ; < 3 * 2^29)
    const Fq29 x3 = norm29(sub29<7, 31>(rr, t));                           // (9 ; 29)
    const Fq29 v = sub29<10, 29>(q, x3);                                   // (12 ; 30.6)
    acc.y = mul2add29<FqParams, SER>(r, v, s1, sub29<3, 29>(zero29(), ppp));              // 5 * 12 + 2 * 3 = 66: (2 ; 29)
    acc.x = x3;
    acc.zz = mul29<FqParams, SER>(mul29<FqParams, SER>(acc.zz, b.zz), pp);
    acc.zzz = mul29<FqParams, SER>(mul29<FqParams, SER>(acc.zzz, b.zzz), ppp);
}

// The point held by lane (lane + off) mod 64.  One source address for all 37 words and raw ds_bpermute, so that the
// 37 transfers are in flight together: through __shfl_down (index arithmetic, clamp and a wait per word) a tree step's
// shuffle cost a lone wave as much as the addition itself (5.7 us against 6.3, tools/ubench_tail.hip).
__device__ __forceinline__ G1X29 g1x29_shfl_down(const G1X29& v, int off) {
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int addr = ((lane + off) & 63) << 2;
    G1X29 r;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        r.x.l[k] = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)v.x.l[k]);
        r.y.l[k] = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)v.y.l[k]);
        r.zz.l[k] = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)v.zz.l[k]);
        r.zzz.l[k] = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)v.zzz.l[k]);
    }
    r.inf = __builtin_amdgcn_ds_bpermute(addr, (int)v.inf) != 0;
    return r;
}

// out-of-line conversion for the one lane that hands a bit sum to the host
__device__ __noinline__ Words8 internal_to_std_vcall(Limbs9 a) {
    Fq29 x;
#pragma unroll
    for (int i = 0; i < 9; i++) x.l[i] = a[i];
    Fq r = from29(mul29(x, const_pow2_29<256, FqParams>()));
    reduce_once(r);
    Words8 o;
#pragma unroll
    for (int i = 0; i < 8; i++) o[i] = r.v[i];
    return o;
}
__device__ __forceinline__ Fq internal_to_std_call(const Fq29& a) {
    Limbs9 x;
#pragma unroll
    for (int i = 0; i < 9; i++) x[i] = a.l[i];
    const Words8 o = internal_to_std_vcall(x);
    Fq r;
#pragma unroll
    for (int i = 0; i < 8; i++) r.v[i] = o[i];
    return r;
}

}  // namespace zk
