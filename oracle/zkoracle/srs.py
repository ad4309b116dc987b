"""KZG SRS of the reference (`gen_srs(k)`), reproduced from the recovered tau.

Oracle (test infrastructure).  Restates halo2_proofs
`poly::kzg::commitment::ParamsKZG::setup(k, ChaCha20Rng::from_seed([0;32]))`
as called through halo2-base `gen_srs` at reference
halo2-circuits/src/ecc/ecdsa_p256.rs:258,279,338,388,430,450,515.
Known answer K1: [tau]G2 == s_g2 in proving-server/P256Verifier.yul:1131-1134
(the verifier stores -[tau]G2).
"""
from .field import R, P, omega, inv, batch_inv
from .hashes import ChaCha20Rng
from . import curve


def tau():
    return ChaCha20Rng(bytes(32)).fr()


TAU = tau()


def lagrange_at(k, x):
    """[L_0(x) .. L_{n-1}(x)] over the 2^k domain; L_i(x) = w^i (x^n - 1) / (n (x - w^i))."""
    n = 1 << k
    w = omega(k)
    xn1 = (pow(x, n, R) - 1) % R
    ninv = inv(n, R)
    ws = [1] * n
    for i in range(1, n):
        ws[i] = ws[i - 1] * w % R
    den = batch_inv([(x - wi) % R for wi in ws], R)
    c = xn1 * ninv % R
    return [ws[i] * c % R * den[i] % R for i in range(n)]


def commit_scalar_monomial(coeffs, x=None):
    """f(tau) for coefficient-form f — commit(f) = [f(tau)] G1."""
    x = TAU if x is None else x
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % R
    return acc


def commit_scalar_lagrange(k, values, lag=None):
    lag = lagrange_at(k, TAU) if lag is None else lag
    return sum(v * l for v, l in zip(values, lag)) % R


def g1_of_scalar(s):
    return curve.mul(curve.G1_GEN, s % R)


def srs_points(k, lagrange=False, count=None):
    """First `count` points of g (monomial) or g_lagrange — slow (one scalar mul
    each); for small k / spot checks only."""
    n = 1 << k
    count = n if count is None else count
    if lagrange:
        sc = lagrange_at(k, TAU)[:count]
    else:
        sc = [pow(TAU, i, R) for i in range(count)]
    return [g1_of_scalar(s) for s in sc]
