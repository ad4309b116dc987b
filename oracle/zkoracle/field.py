"""BN254 base/scalar field constants and helpers (plain Python ints).

Oracle (test infrastructure) — see oracle/zkoracle/__init__.py.
Restates halo2curves `bn256::{Fq,Fr}` (4x64-limb Montgomery, R = 2^256);
moduli as in reference proving-server/P256Verifier.yul:17-18.
"""

P = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47  # Fq (base field)
R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001  # Fr (scalar field)

MONT_R = 1 << 256
S = 28  # 2-adicity of Fr
GENERATOR = 7  # multiplicative generator of Fr (halo2curves MULTIPLICATIVE_GENERATOR)
ROOT_OF_UNITY = pow(GENERATOR, (R - 1) >> S, R)  # primitive 2^28-th root
DELTA = pow(GENERATOR, 1 << S, R)  # generator of the odd-order subgroup (permutation cosets)
# ZETA: primitive cube root of unity used by halo2's extended-domain coset ("g_coset" = Fr::ZETA).  Either cube
# root gives the same h(X), hence the same proof bytes; the extended cosets stored in a ProvingKey file differ, so
# this is halo2curves' constant [RECALLED: bn256 Fr::ZETA; checked below to be (7^((r-1)/3))^2, a primitive cube root].
ZETA = 0x30644E72E131A029048B6E193FD84104CC37A73FEC2BC5E9B8CA0B2D36636F23
assert ZETA == pow(pow(GENERATOR, (R - 1) // 3, R), 2, R) and pow(ZETA, 3, R) == 1 and ZETA != 1


def inv(a, m):
    return pow(a, -1, m)


def to_mont(a, m):
    return (a * MONT_R) % m


def from_mont(a, m):
    return (a * inv(MONT_R, m)) % m


def omega(k):
    """Primitive 2^k-th root of unity: ROOT_OF_UNITY^(2^(S-k))."""
    return pow(ROOT_OF_UNITY, 1 << (S - k), R)


def batch_inv(xs, m):
    """Montgomery's trick; zeros map to zero (as halo2's batch_invert)."""
    n = len(xs)
    pref = [1] * (n + 1)
    for i, x in enumerate(xs):
        pref[i + 1] = pref[i] * (x if x else 1) % m
    acc = inv(pref[n], m)
    out = [0] * n
    for i in range(n - 1, -1, -1):
        x = xs[i]
        if x:
            out[i] = acc * pref[i] % m
            acc = acc * x % m
    return out


def limbs_le(a, n=4):
    return [(a >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)]


def from_limbs_le(l):
    return sum(int(x) << (64 * i) for i, x in enumerate(l))
