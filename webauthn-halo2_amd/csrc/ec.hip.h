// ec.hip.h — BN254 G1 (y^2 = x^3 + 3) point arithmetic for gfx950 device code.
//
// Device twin of halo2curves `bn256::{G1Affine, G1}` as used by
// halo2_proofs `best_multiexp` (reference call sites
// halo2-circuits/src/ecc/ecdsa_p256.rs:366-373,416-423).  Accumulators use
// extended-Jacobian XYZZ coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2):
// a mixed add is 8M + 2S with no inversion, and identity is ZZ = 0.
#pragma once
#include "field.hip.h"

namespace zk {

struct alignas(16) G1Affine {  // memory image of halo2curves G1Affine; identity = (0,0)
    Fq x, y;
};

struct alignas(16) G1Jac {  // memory image of halo2curves G1 (Jacobian); identity z = 0
    Fq x, y, z;
};

struct alignas(16) G1X {  // XYZZ accumulator
    Fq x, y, zz, zzz;

    __host__ __device__ __forceinline__ static G1X identity() {
        G1X r;
        r.x = Fq::one();
        r.y = Fq::one();
        r.zz = Fq::zero();
        r.zzz = Fq::zero();
        return r;
    }
    __host__ __device__ __forceinline__ bool is_identity() const { return zz.is_zero(); }
};

__host__ __device__ __forceinline__ bool affine_is_identity(const G1Affine& p) { return p.x.is_zero() && p.y.is_zero(); }

// XYZZ doubling (dbl-2008-s-1, a = 0)
__host__ __device__ inline G1X g1x_dbl(const G1X& p) {
    if (p.is_identity()) return p;
    Fq u = fe_dbl(p.y);
    Fq v = fe_sqr(u);
    Fq w = fe_mul(u, v);
    Fq s = fe_mul(p.x, v);
    Fq xx = fe_sqr(p.x);
    Fq m = fe_add(fe_dbl(xx), xx);
    G1X r;
    r.x = fe_sub(fe_sqr(m), fe_dbl(s));
    r.y = fe_sub(fe_mul(m, fe_sub(s, r.x)), fe_mul(w, p.y));
    r.zz = fe_mul(v, p.zz);
    r.zzz = fe_mul(w, p.zzz);
    return r;
}

__host__ __device__ inline G1X g1x_dbl_affine(const Fq& x, const Fq& y) {
    // doubling of an affine point (ZZ = ZZZ = 1): mdbl-2008-s-1
    Fq u = fe_dbl(y);
    Fq v = fe_sqr(u);
    Fq w = fe_mul(u, v);
    Fq s = fe_mul(x, v);
    Fq xx = fe_sqr(x);
    Fq m = fe_add(fe_dbl(xx), xx);
    G1X r;
    r.x = fe_sub(fe_sqr(m), fe_dbl(s));
    r.y = fe_sub(fe_mul(m, fe_sub(s, r.x)), fe_mul(w, y));
    r.zz = v;
    r.zzz = w;
    return r;
}

// acc += (x, y) affine (madd-2008-s); (x, y) must not be the identity.
__host__ __device__ inline void g1x_add_affine(G1X& acc, const Fq& x, const Fq& y) {
    if (acc.is_identity()) {
        acc.x = x;
        acc.y = y;
        acc.zz = Fq::one();
        acc.zzz = Fq::one();
        return;
    }
    Fq u2 = fe_mul(x, acc.zz);
    Fq s2 = fe_mul(y, acc.zzz);
    Fq p = fe_sub(u2, acc.x);
    Fq r = fe_sub(s2, acc.y);
    if (p.is_zero()) {
        if (r.is_zero()) acc = g1x_dbl_affine(x, y);
        else acc = G1X::identity();
        return;
    }
    Fq pp = fe_sqr(p);
    Fq ppp = fe_mul(p, pp);
    Fq q = fe_mul(acc.x, pp);
    Fq x3 = fe_sub(fe_sub(fe_sqr(r), ppp), fe_dbl(q));
    Fq y3 = fe_sub(fe_mul(r, fe_sub(q, x3)), fe_mul(acc.y, ppp));
    acc.x = x3;
    acc.y = y3;
    acc.zz = fe_mul(acc.zz, pp);
    acc.zzz = fe_mul(acc.zzz, ppp);
}

// acc += b (add-2008-s), both XYZZ, all special cases handled.
__host__ __device__ inline void g1x_add(G1X& acc, const G1X& b) {
    if (b.is_identity()) return;
    if (acc.is_identity()) {
        acc = b;
        return;
    }
    Fq u1 = fe_mul(acc.x, b.zz);
    Fq u2 = fe_mul(b.x, acc.zz);
    Fq s1 = fe_mul(acc.y, b.zzz);
    Fq s2 = fe_mul(b.y, acc.zzz);
    Fq p = fe_sub(u2, u1);
    Fq r = fe_sub(s2, s1);
    if (p.is_zero()) {
        if (r.is_zero()) acc = g1x_dbl(acc);
        else acc = G1X::identity();
        return;
    }
    Fq pp = fe_sqr(p);
    Fq ppp = fe_mul(p, pp);
    Fq q = fe_mul(u1, pp);
    Fq x3 = fe_sub(fe_sub(fe_sqr(r), ppp), fe_dbl(q));
    Fq y3 = fe_sub(fe_mul(r, fe_sub(q, x3)), fe_mul(s1, ppp));
    acc.x = x3;
    acc.y = y3;
    acc.zz = fe_mul(fe_mul(acc.zz, b.zz), pp);
    acc.zzz = fe_mul(fe_mul(acc.zzz, b.zzz), ppp);
}

// XYZZ -> Jacobian with Z = ZZZ/ZZ would need an inversion; instead use the
// isomorphic representative (X*ZZ^2... ) : (X', Y', Z') = (X*ZZ, Y*ZZZ, ZZ)
// satisfies X'/Z'^2 = X/ZZ and Y'/Z'^3 = Y*ZZZ/ZZ^3 = Y/ZZZ (since ZZ^3 = ZZZ^2).
__host__ __device__ inline G1Jac g1x_to_jac(const G1X& p) {
    G1Jac r;
    if (p.is_identity()) {
        r.x = Fq::one();
        r.y = Fq::one();
        r.z = Fq::zero();
        return r;
    }
    r.x = fe_mul(p.x, p.zz);
    r.y = fe_mul(p.y, p.zzz);
    r.z = p.zz;
    return r;
}

__device__ __forceinline__ G1Affine affine_load(const G1Affine* p) {
    G1Affine r;
    r.x = fe_load(&p->x);
    r.y = fe_load(&p->y);
    return r;
}

__device__ __forceinline__ G1X g1x_load(const G1X* p) {
    G1X r;
    r.x = fe_load(&p->x);
    r.y = fe_load(&p->y);
    r.zz = fe_load(&p->zz);
    r.zzz = fe_load(&p->zzz);
    return r;
}

__device__ __forceinline__ void g1x_store(G1X* p, const G1X& a) {
    fe_store(&p->x, a.x);
    fe_store(&p->y, a.y);
    fe_store(&p->zz, a.zz);
    fe_store(&p->zzz, a.zzz);
}

}  // namespace zk
