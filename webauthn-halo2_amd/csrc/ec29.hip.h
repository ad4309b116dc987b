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
// or a cancellation), which the caller redoes on the general path; acc is then unchanged.
__device__ __forceinline__ bool g1x29_add_affine(G1X29& acc, const Fq& x, const Fq& y) {
    if (acc.inf) {
        acc.x = std_to_internal(x);  // (2 ; 29)
        acc.y = std_to_internal(y);
        acc.zz = const_pow2_29<261, FqParams>();  // 1 in internal form
        acc.zzz = acc.zz;
        acc.inf = false;
        return true;
    }
    const Fq29 x2 = to29_x32(x), y2 = to29_x32(y);               // (32 ; 29)
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
    if (is_zero29(pp)) return false;                              // p prime: P = 0 (mod p) <=> P^2 = 0
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

}  // namespace zk
