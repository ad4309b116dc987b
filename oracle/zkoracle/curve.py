"""BN254 G1 (y^2 = x^3 + 3) and the little of G2 needed for the K1 check.

Oracle (test infrastructure).  Restates halo2curves `bn256::{G1,G1Affine}`;
curve equation as checked by reference proving-server/P256Verifier.yul:25-31.
Points are affine tuples (x, y) of ints, identity = None.
"""
from .field import P, inv

B = 3
G1_GEN = (1, 2)  # reference P256Verifier.yul:777-778; ecdsa_p256.rs:290


def is_on_curve(pt):
    if pt is None:
        return True
    x, y = pt
    return x < P and y < P and (y * y - x * x * x - B) % P == 0


def neg(pt):
    if pt is None:
        return None
    return (pt[0], (-pt[1]) % P)


# ---- Jacobian arithmetic (X, Y, Z), identity Z == 0 -------------------------

def _jdbl(p):
    X, Y, Z = p
    if Z == 0 or Y == 0:
        return (1, 1, 0)
    A = X * X % P
    Bq = Y * Y % P
    C = Bq * Bq % P
    D = 2 * ((X + Bq) * (X + Bq) - A - C) % P
    E = 3 * A % P
    F = E * E % P
    X3 = (F - 2 * D) % P
    Y3 = (E * (D - X3) - 8 * C) % P
    Z3 = 2 * Y * Z % P
    return (X3, Y3, Z3)


def _jadd(p, q):
    X1, Y1, Z1 = p
    X2, Y2, Z2 = q
    if Z1 == 0:
        return q
    if Z2 == 0:
        return p
    Z1Z1 = Z1 * Z1 % P
    Z2Z2 = Z2 * Z2 % P
    U1 = X1 * Z2Z2 % P
    U2 = X2 * Z1Z1 % P
    S1 = Y1 * Z2 * Z2Z2 % P
    S2 = Y2 * Z1 * Z1Z1 % P
    if U1 == U2:
        if S1 == S2:
            return _jdbl(p)
        return (1, 1, 0)
    H = (U2 - U1) % P
    I = 4 * H * H % P
    J = H * I % P
    r = 2 * (S2 - S1) % P
    V = U1 * I % P
    X3 = (r * r - J - 2 * V) % P
    Y3 = (r * (V - X3) - 2 * S1 * J) % P
    Z3 = ((Z1 + Z2) * (Z1 + Z2) - Z1Z1 - Z2Z2) * H % P
    return (X3, Y3, Z3)


def to_jac(pt):
    return (1, 1, 0) if pt is None else (pt[0], pt[1], 1)


def to_affine(j):
    X, Y, Z = j
    if Z == 0:
        return None
    zi = inv(Z, P)
    zi2 = zi * zi % P
    return (X * zi2 % P, Y * zi2 * zi % P)


def add(a, b):
    return to_affine(_jadd(to_jac(a), to_jac(b)))


def mul(pt, k):
    """[k]pt, double-and-add, k any non-negative int."""
    acc = (1, 1, 0)
    base = to_jac(pt)
    while k:
        if k & 1:
            acc = _jadd(acc, base)
        base = _jdbl(base)
        k >>= 1
    return to_affine(acc)


def msm_naive(scalars, points):
    acc = (1, 1, 0)
    for s, pt in zip(scalars, points):
        if s and pt is not None:
            acc = _jadd(acc, to_jac(mul(pt, s)))
    return to_affine(acc)


# ---- G2 over Fq2 = Fq[u]/(u^2+1), only for the K1 known-answer check --------

def f2mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def f2add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def f2sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def f2inv(a):
    d = inv((a[0] * a[0] + a[1] * a[1]) % P, P)
    return (a[0] * d % P, (-a[1]) * d % P)


# EVM encoding order is (x.c1, x.c0, y.c1, y.c0) — reference P256Verifier.yul:1125-1128
G2_GEN = (
    (0x1800DEEF121F1E76426A00665E5C4479674322D4F75EDADD46DEBD5CD992F6ED,
     0x198E9393920D483A7260BFB731FB5D25F1AA493335A9E71297E485B7AEF312C2),
    (0x12C85EA5DB8C6DEB4AAB71808DCB408FE3D1E7690C43D37B4CE6CC0166FA7DAA,
     0x090689D0585FF075EC9E99AD690C3395BC4B313370B38EF355ACDADCD122975B),
)


def g2_add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    (x1, y1), (x2, y2) = a, b
    if x1 == x2:
        if y1 == y2:
            lam = f2mul(f2mul((3, 0), f2mul(x1, x1)), f2inv(f2mul((2, 0), y1)))
        else:
            return None
    else:
        lam = f2mul(f2sub(y2, y1), f2inv(f2sub(x2, x1)))
    x3 = f2sub(f2sub(f2mul(lam, lam), x1), x2)
    y3 = f2sub(f2mul(lam, f2sub(x1, x3)), y1)
    return (x3, y3)


def g2_mul(pt, k):
    acc = None
    while k:
        if k & 1:
            acc = g2_add(acc, pt)
        pt = g2_add(pt, pt)
        k >>= 1
    return acc
