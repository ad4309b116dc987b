"""GPU parity: HIP MSM / NTT / SRS / eval through the C-ABI vs the oracle.

Bit-exact (integer arithmetic; group elements compared after affine
normalisation, which is canonical).  Sizes the oracle finishes in seconds, plus
size-independent properties at BASELINE sizes (2^19, 2^21)."""
import random

import numpy as np
import pytest

from zkoracle import cops, curve as C, field as F, srs

pytestmark = pytest.mark.gpu


def rand_fr(rng, n):
    return [rng.randrange(F.R) for _ in range(n)]


def mont1(x):
    return cops.fr_mont([x])[0]


# ------------------------------------------------------------------- NTT ----

@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 5, 7, 8, 9, 11, 13, 14, 15, 17])
def test_ntt_matches_oracle(engine, log_n):
    n = 1 << log_n
    a = np.frombuffer(np.random.default_rng(log_n).bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
    a[:, 3] &= 0x0FFFFFFFFFFFFFFF  # < 2^252 < r: valid Montgomery residues
    w = F.omega(log_n)
    want = cops.ntt(a, w, log_n)
    got = engine.ntt(a, mont1(w), log_n)
    assert np.array_equal(got, want)
    # inverse root takes the reversed-table path
    wi = F.inv(w, F.R)
    assert np.array_equal(engine.ntt(a, mont1(wi), log_n), cops.ntt(a, wi, log_n))


@pytest.mark.parametrize("log_n", [9, 14, 17, 19])
def test_ntt_largest_stored_words(engine, log_n):
    """The butterflies run lazily on 29-bit limbs (value bounds grow by stage, ntt.hip): the worst operands are the largest
    stored words.  Every element p - 1 (as stored), and p - 1 alternating with 0, through one, two and three passes."""
    n = 1 << log_n
    pm1 = np.array([(F.R - 1) >> (64 * i) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
    w = F.omega(log_n)
    for pattern in ("all", "alt"):
        a = np.tile(pm1, (n, 1))
        if pattern == "alt":
            a[1::2] = 0
        assert np.array_equal(engine.ntt(a, mont1(w), log_n), cops.ntt(a, w, log_n)), (log_n, pattern)


def test_ntt_nonstandard_omega(engine):
    # a 2^6-th root used on a 2^6 domain but not the canonical one (odd power): own twiddle table
    log_n = 6
    w = pow(F.omega(log_n), 5, F.R)
    a = cops.fr_mont(rand_fr(random.Random(5), 1 << log_n))
    assert np.array_equal(engine.ntt(a, mont1(w), log_n), cops.ntt(a, w, log_n))


@pytest.mark.parametrize("log_n", [19, 21])
def test_ntt_roundtrip_large(engine, log_n):
    n = 1 << log_n
    a = np.frombuffer(np.random.default_rng(77).bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
    a[:, 3] &= 0x0FFFFFFFFFFFFFFF
    w = F.omega(log_n)
    fwd = engine.ntt(a, mont1(w), log_n)
    # spot-check 3 outputs against the definition via Horner evaluation at w^i (oracle side)
    ai = cops.fr_ints(a)
    for i in (0, 1, n - 1):
        x = pow(w, i, F.R)
        acc = 0
        for c in reversed(ai):
            acc = (acc * x + c) % F.R
        assert cops.fr_ints(fwd[i:i + 1])[0] == acc
    back = engine.ntt(fwd, mont1(F.inv(w, F.R)), log_n)
    ninv = cops.fr_mont([F.inv(n, F.R)])
    # back = n * a  ->  compare a few hundred entries exactly, and all via linear check
    bi = cops.fr_ints(back[:256])
    assert [x * F.inv(n, F.R) % F.R for x in bi] == ai[:256]
    p = engine.poly(n, fwd)
    engine.lagrange_to_coeff(p)  # fused 1/n
    assert np.array_equal(engine.download(p), a)
    p.free()
    del ninv


# ------------------------------------------------------------------- MSM ----

def small_srs(n):
    return cops.fixed_base_g1(cops.fr_powers(srs.TAU, n))


@pytest.mark.parametrize("n", [0, 1, 2, 3, 17, 100, 1000, 4096, 5000, 1 << 14])
def test_msm_matches_oracle(engine, n):
    rng = random.Random(n)
    g = small_srs(max(n, 1))[:n]
    s = rand_fr(rng, n)
    if n >= 17:
        s[0] = 0
        s[1] = 1
        s[2] = F.R - 1
        s[3] = 2
        s[4] = (1 << 128) - 1
        s[5] = 1 << 253
    sm = cops.fr_mont(s) if n else np.zeros((0, 4), dtype=np.uint64)
    got = cops.jac_to_affine_ints(engine.msm(sm, g))
    want = cops.jac_to_affine_ints(cops.msm(sm, g)) if n else None
    assert got == want


def test_msm_edge_bases(engine):
    """identity bases, repeated bases (doubling path), P and -P (cancellation)."""
    rng = random.Random(99)
    n = 2048
    g = small_srs(n)
    g[10] = 0  # identity
    g[11] = 0
    g[20] = g[21]  # same point twice
    g[22] = g[21]
    neg = cops.affine_arr_to_ints(g[30:31])[0]
    negp = cops.to_mont_arr(cops.ints_to_arr([neg[0], (-neg[1]) % F.P]), 1).reshape(8)
    g[31] = negp  # g[31] = -g[30]
    s = rand_fr(rng, n)
    s[20] = s[21] = s[22] = 5  # equal scalars on equal points -> same bucket -> doubling
    s[30] = s[31] = 7  # P and -P in one bucket -> identity
    sm = cops.fr_mont(s)
    assert cops.jac_to_affine_ints(engine.msm(sm, g)) == cops.jac_to_affine_ints(cops.msm(sm, g))
    # all-zero scalars -> identity
    z = np.zeros((n, 4), dtype=np.uint64)
    assert cops.jac_to_affine_ints(engine.msm(z, g)) is None
    # all scalars equal one -> sum of points
    ones = cops.fr_mont([1] * n)
    assert cops.jac_to_affine_ints(engine.msm(ones, g)) == cops.jac_to_affine_ints(cops.msm(ones, g))


@pytest.mark.parametrize("with_identity", [False, True])
def test_fixed_base_commit_over_degenerate_srs(engine, with_identity):
    """The fixed-base (window-table) path over an SRS that holds the same point several times and a P / -P pair: equal
    scalars put them in one bucket (doubling, cancellation).  A basis without the identity takes the unchecked accumulation
    loop, which must notice the exceptional step (ZZ = 0 at the end of the segment) and redo the segment on the general
    formulas; a basis with an identity point takes the checked loop.  Both bases (coefficient and Lagrange) are loaded."""
    k = 10
    n = 1 << k
    rng = random.Random(1234)
    g = small_srs(n)
    g[20] = g[21]
    g[22] = g[21]
    g[23] = g[21]
    for i in range(40, 60):   # a run of equal points: many doublings in one segment
        g[i] = g[40]
    neg = cops.affine_arr_to_ints(g[30:31])[0]
    g[31] = cops.to_mont_arr(cops.ints_to_arr([neg[0], (-neg[1]) % F.P]), 1).reshape(8)
    if with_identity:
        g[10] = 0
        g[700] = 0
    gl = g[::-1].copy()
    engine.srs_load(k, g, gl)
    s = rand_fr(rng, n)
    s[20] = s[21] = s[22] = s[23] = 5
    for i in range(40, 60):
        s[i] = 9
    s[30] = s[31] = 7
    sm = cops.fr_mont(s)
    p = engine.poly(n, sm)
    for basis, arr_ in ((0, g), (1, gl)):
        got = cops.affine_arr_to_ints(engine.commit(p, basis))[0]
        assert got == cops.jac_to_affine_ints(cops.msm(sm, arr_)), (with_identity, basis)
    ones = engine.poly(n, cops.fr_mont([1] * n))
    assert cops.affine_arr_to_ints(engine.commit(ones, 0))[0] == cops.jac_to_affine_ints(cops.msm(cops.fr_mont([1] * n), g))
    p.free()
    ones.free()


def test_msm_witness_like_distribution(engine):
    """Hot low buckets: the mix SURVEY.md §8d prescribes for advice columns."""
    rng = random.Random(0x5EED0019)
    n = 1 << 15
    g = small_srs(n)
    s = []
    for _ in range(n):
        u = rng.random()
        if u < 0.40:
            s.append(rng.randrange(1 << 18))
        elif u < 0.75:
            s.append(rng.randrange(1 << 88))
        elif u < 0.90:
            s.append(rng.randrange(F.R))
        else:
            s.append(0)
    sm = cops.fr_mont(s)
    assert cops.jac_to_affine_ints(engine.msm(sm, g)) == cops.jac_to_affine_ints(cops.msm(sm, g))
    b = cops.fr_mont([rng.randrange(2) for _ in range(n)])  # boolean column: one giant bucket
    assert cops.jac_to_affine_ints(engine.msm(b, g)) == cops.jac_to_affine_ints(cops.msm(b, g))


# ------------------------------------------------------------- SRS / commit ---

def test_srs_setup_matches_reference_known_answers(engine):
    k = 17
    n = 1 << k
    engine.srs_setup(k)  # seed [0;32] = gen_srs
    g = cops.affine_arr_to_ints(engine.srs_export(0, 0, 4))
    assert g[0] == C.G1_GEN  # reference P256Verifier.yul:777-778
    assert g[1] == C.mul(C.G1_GEN, srs.TAU)
    assert g[3] == C.mul(C.G1_GEN, pow(srs.TAU, 3, F.R))
    last = cops.affine_arr_to_ints(engine.srs_export(0, n - 1, 1))[0]
    assert last == C.mul(C.G1_GEN, pow(srs.TAU, n - 1, F.R))
    lag = srs.lagrange_at(k, srs.TAU)
    gl = cops.affine_arr_to_ints(engine.srs_export(1, 0, 2)) + cops.affine_arr_to_ints(engine.srs_export(1, n - 2, 2))
    for pt, i in zip(gl, (0, 1, n - 2, n - 1)):
        assert pt == C.mul(C.G1_GEN, lag[i])
    # K2: commit_lagrange(range table 0..2^16-1) == fixed-column commitment of the reference's
    # k=17 verifying key, proving-server/P256Verifier.yul:889-890
    table = cops.fr_mont(list(range(1 << 16)) + [0] * (n - (1 << 16)))
    p = engine.poly(n, table)
    c = cops.affine_arr_to_ints(engine.commit(p, 1))[0]
    assert c == (
        0x2F579160607CC547A54EF72E5A1A2966A65305C955CF8D94F507169386A10F4C,
        0x15932D491AAAA6D3673EEB19941A96EE53B011A6923028A70466A155B753D46B,
    )
    # commit(iNTT(v)) == commit_lagrange(v)
    engine.lagrange_to_coeff(p)
    assert cops.affine_arr_to_ints(engine.commit(p, 0))[0] == c
    p.free()


@pytest.mark.parametrize("k", [18, 19])
def test_commit_tau_oracle_k19(engine, k):
    """MSM(s, SRS) == [sum s_i tau^i] G1 at the BASELINE size 2^19 (any-n oracle); 2^18 is the shortest column that
    takes the two-level counting sort (msm.hip sort2_applies)."""
    n = 1 << k
    engine.srs_setup(k)
    a = np.frombuffer(np.random.default_rng(0x5EED0019).bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
    a[:, 3] &= 0x0FFFFFFFFFFFFFFF
    p = engine.poly(n, a)
    got = cops.affine_arr_to_ints(engine.commit(p, 0))[0]
    want = srs.g1_of_scalar(srs.commit_scalar_monomial(cops.fr_ints(a)))
    assert got == want
    # and through the fine-grained seam with bases exported back to the host
    bases = engine.srs_export(0, 0, n)
    assert cops.jac_to_affine_ints(engine.msm(a, bases)) == want
    p.free()


@pytest.mark.parametrize("k,count", [(10, 70), (14, 9), (17, 5), (19, 3)])
def test_commit_batch_equals_single_commits(engine, k, count):
    """Column-batched MSM passes (how the prover commits advice columns, grand products, h pieces): every
    result equals the one-column commitment, both bases; columns include zeros, a sparse one and full ones;
    `count` exceeds the per-pass batch at k = 10 and k = 19 (several passes)."""
    n = 1 << k
    engine.srs_setup(k)
    rng = np.random.default_rng(k * 1000 + count)
    polys = []
    for j in range(count):
        a = np.frombuffer(rng.bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
        a[:, 3] &= 0x0FFFFFFFFFFFFFFF
        if j == 1:
            a[:] = 0                      # the zero column commits to the identity
        if j == 2:
            a[:, 1:] = 0
            a[:, 0] &= 0xFFFF             # small values: a few hot buckets
            a[::3] = 0
        polys.append(engine.poly(n, a))
    for basis in (0, 1):
        got = engine.commit_batch(polys, basis)
        for j, p in enumerate(polys):
            assert (got[j] == engine.commit(p, basis)).all(), (k, basis, j)
        assert not got[1].any()
    for p in polys:
        p.free()


# ------------------------------------------------------------ eval / coset ----

@pytest.mark.parametrize("log_n", [0, 1, 3, 8, 11, 12, 16, 19, 22])
def test_kate_division_matches_oracle(engine, log_n):
    """arithmetic::kate_division through zk_kate_division, all coefficients against the oracle's restatement; 2^22 is the
    first size whose chunk count exceeds one top-level workgroup at the shortest chunk (the chunk length grows there)."""
    from zkoracle import fastprover as fp

    n = 1 << log_n
    a = np.frombuffer(np.random.default_rng(900 + log_n).bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
    a[:, 3] &= 0x0FFFFFFFFFFFFFFF
    z = random.Random(log_n).randrange(F.R)
    want = fp.kate_division(a, z)
    p = engine.poly(n, a)
    q = engine.poly(n)
    engine.kate_division(p, mont1(z), q)
    got = engine.download(q)
    assert np.array_equal(got[:n - 1], want[:n - 1]) and not got[n - 1].any()
    engine.kate_division(p, mont1(z))  # in place
    assert np.array_equal(engine.download(p), got)
    p.free()
    q.free()


@pytest.mark.parametrize("n", [1, 2, 255, 256, 257, 4096, 70000])
def test_eval_matches_horner(engine, n):
    rng = random.Random(n)
    c = rand_fr(rng, n)
    x = rng.randrange(F.R)
    p = engine.poly(n, cops.fr_mont(c))
    got = cops.fr_ints(engine.eval(p, mont1(x)).reshape(1, 4))[0]
    acc = 0
    for ci in reversed(c):
        acc = (acc * x + ci) % F.R
    assert got == acc
    p.free()


@pytest.mark.parametrize("k", [4, 9, 12])
def test_coset_extended_domain(engine, k):
    """coeff_to_extended: ext[i] = f(zeta * w_ext^i); extended_to_coeff inverts it."""
    rng = random.Random(k)
    n, ext = 1 << k, 1 << (k + 2)
    f = rand_fr(rng, n)
    src = engine.poly(n, cops.fr_mont(f))
    dst = engine.poly(ext)
    engine.coeff_to_extended(src, dst)
    got = cops.fr_ints(engine.download(dst))
    wext = F.omega(k + 2)
    for i in (0, 1, 2, 3, ext // 2 + 1, ext - 1):
        x = F.ZETA * pow(wext, i, F.R) % F.R
        acc = 0
        for c in reversed(f):
            acc = (acc * x + c) % F.R
        assert got[i] == acc
    engine.extended_to_coeff(dst, ext)
    back = cops.fr_ints(engine.download(dst))
    assert back[:n] == f and all(v == 0 for v in back[n:])
    src.free()
    dst.free()


def test_k21_stress_msm_tau_oracle(engine):
    """BASELINE configs[4]: 2^21-point MSM against the device SRS == [sum s_i tau^i] G1."""
    k = 21
    n = 1 << k
    engine.srs_setup(k)
    a = np.frombuffer(np.random.default_rng(0x5EED0021).bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
    a[:, 3] &= 0x0FFFFFFFFFFFFFFF
    p = engine.poly(n, a)
    got = cops.affine_arr_to_ints(engine.commit(p, 0))[0]
    assert got == srs.g1_of_scalar(srs.commit_scalar_monomial(cops.fr_ints(a)))
    # linearity (size-independent property): commit(2 a) == commit(a) + commit(a)
    q = engine.poly(n, cops.fr_mont([x * 2 % F.R for x in cops.fr_ints(a)]))
    assert cops.affine_arr_to_ints(engine.commit(q, 0))[0] == C.add(got, got)
    q.free()
    p.free()
    engine.srs_setup(12)  # release the 5 GB of k=21 tables for the tests that follow
