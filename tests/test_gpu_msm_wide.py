"""GPU parity for the wide MSM path (15 / 16-bit windows over the resident SRS: msm.hip "wide path"): every commitment
equals the oracle's (tau-oracle / C Pippenger), bit-exact after affine normalisation.  The path is selected with
ZK_OPT_MSM_WINDOW before the SRS (and its window tables) is loaded; 16 bits is the default at k = 19."""
import random

import numpy as np
import pytest

from zkoracle import cops, curve as C, field as F, srs

pytestmark = pytest.mark.gpu


@pytest.fixture()
def windowed(engine):
    from webauthn_halo2_amd import engine as E

    def select(bits):
        engine.set_option(E.ZK_OPT_MSM_WINDOW, bits)
        return engine

    yield select
    engine.set_option(E.ZK_OPT_MSM_WINDOW, 0)
    engine.srs_setup(10)  # tables of the default plan for whoever comes next


def rand_col(rng, n):
    a = np.frombuffer(rng.bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
    a[:, 3] &= 0x0FFFFFFFFFFFFFFF
    return a


def tau_commit(a):
    return srs.g1_of_scalar(srs.commit_scalar_monomial(cops.fr_ints(a)))


@pytest.mark.parametrize("bits", [15, 16, 17])
@pytest.mark.parametrize("k", [10, 13, 16])
def test_wide_commit_matches_tau_oracle(windowed, bits, k):
    eng = windowed(bits)
    n = 1 << k
    eng.srs_setup(k)
    assert eng.srs_msm_plan()[0] == bits
    rng = np.random.default_rng(1000 * bits + k)
    cols = [rand_col(rng, n) for _ in range(5)]
    cols[1][:] = 0                                   # the zero column: the identity
    cols[2][:, 1:] = 0                               # 16-bit values: one window, a few hot buckets
    cols[2][:, 0] &= 0xFFFF
    cols[2][::3] = 0
    pr = random.Random(k)
    mix = []                                         # the advice-column mix of SURVEY.md 8d
    for _ in range(n):
        u = pr.random()
        mix.append(pr.randrange(1 << 18) if u < 0.4 else pr.randrange(1 << 88) if u < 0.75 else pr.randrange(F.R) if u < 0.9 else 0)
    mixm = cops.fr_mont(mix)
    boolc = cops.fr_mont([pr.randrange(2) for _ in range(n)])   # one giant bucket
    top = cops.fr_mont([F.R - 1 - pr.randrange(3) for _ in range(n)])  # largest scalars: every window's top digits, carries
    # a handful of distinct digits: non-empty buckets hundreds apart and whole coarse bins empty — every first-of-bucket entry
    # takes the escape of the lanes' bucket tracking (distance field saturated / first bucket of a bin)
    vals = [1, 40000, (3 << 16) | 7, (12345 << 32) | (65535 << 16) | 200, (1 << 253) | (9 << 128), F.R - 2]
    sparse = cops.fr_mont([vals[pr.randrange(len(vals))] if pr.random() < 0.7 else 0 for _ in range(n)])
    # every digit of every scalar the same: ALL 16 n entries of the column in one bucket (one bin, one bucket, every
    # accumulation lane a slot of it, thousands of parts); and the digit 0x8000 throughout (-2^15 with a carry into every window)
    ones = cops.fr_mont([sum(1 << (bits * w) for w in range(250 // bits))] * n)
    halves = cops.fr_mont([sum((1 << (bits - 1)) << (bits * w) for w in range(250 // bits))] * n)
    polys = [eng.poly(n, c) for c in cols[:3]] + [eng.poly(n, x) for x in (mixm, boolc, top, sparse, ones, halves)]
    data = cols[:3] + [mixm, boolc, top, sparse, ones, halves]
    want = [tau_commit(d) for d in data]
    for j, p in enumerate(polys):
        got = cops.affine_arr_to_ints(eng.commit(p, 0))[0]
        assert got == want[j], (bits, k, j)
    # the same columns through ONE batched pass, both bases
    got = eng.commit_batch(polys, 0)
    for j in range(len(polys)):
        assert cops.affine_arr_to_ints(got[j:j + 1])[0] == want[j], (bits, k, "batch", j)
    gl = eng.commit_batch(polys, 1)
    for j, p in enumerate(polys):
        assert (gl[j] == eng.commit(p, 1)).all()
    # commit_lagrange(v) == commit(iNTT(v))
    eng.lagrange_to_coeff(polys[0])
    assert (eng.commit(polys[0], 0) == gl[0]).all()
    for p in polys:
        p.free()


@pytest.mark.parametrize("bits", [16, 17])
@pytest.mark.parametrize("with_identity", [False, True])
def test_wide_commit_over_degenerate_srs(windowed, with_identity, bits):
    """Equal points with equal scalars (doublings inside a segment), P / -P (cancellation), an SRS with the identity:
    the unchecked accumulation + redo kernel and the checked one, on the wide path's column regions."""
    eng = windowed(bits)
    k = 10
    n = 1 << k
    rng = random.Random(4321)
    g = cops.fixed_base_g1(cops.fr_powers(srs.TAU, n))
    g[20] = g[21]
    g[22] = g[21]
    for i in range(40, 60):
        g[i] = g[40]
    neg = cops.affine_arr_to_ints(g[30:31])[0]
    g[31] = cops.to_mont_arr(cops.ints_to_arr([neg[0], (-neg[1]) % F.P]), 1).reshape(8)
    if with_identity:
        g[10] = 0
        g[700] = 0
    gl = g[::-1].copy()
    eng.srs_load(k, g, gl)
    assert eng.srs_msm_plan()[0] == bits
    s = [rng.randrange(F.R) for _ in range(n)]
    s[20] = s[21] = s[22] = 5
    for i in range(40, 60):
        s[i] = 9
    s[30] = s[31] = 7
    sm = cops.fr_mont(s)
    p = eng.poly(n, sm)
    q = eng.poly(n, cops.fr_mont([1] * n))
    for basis, arr_ in ((0, g), (1, gl)):
        assert cops.affine_arr_to_ints(eng.commit(p, basis))[0] == cops.jac_to_affine_ints(cops.msm(sm, arr_))
        both = eng.commit_batch([p, q], basis)
        assert cops.affine_arr_to_ints(both[0:1])[0] == cops.jac_to_affine_ints(cops.msm(sm, arr_))
        assert cops.affine_arr_to_ints(both[1:2])[0] == cops.jac_to_affine_ints(cops.msm(cops.fr_mont([1] * n), arr_))
    p.free()
    q.free()


@pytest.mark.parametrize("bits", [15, 16, 17])
def test_wide_commit_tau_oracle_k19(windowed, bits):
    """BASELINE size: MSM(s, SRS) == [sum s_i tau^i] G1 at 2^19, one column and a three-column pass."""
    eng = windowed(bits)
    k = 19
    n = 1 << k
    eng.srs_setup(k)
    rng = np.random.default_rng(0x5EED0019 + bits)
    cols = [rand_col(rng, n) for _ in range(3)]
    cols[2][:, 1:] = 0
    cols[2][:, 0] &= 0x3FFFF   # lookup-table-like: 18-bit values
    polys = [eng.poly(n, c) for c in cols]
    want = [tau_commit(c) for c in cols]
    assert cops.affine_arr_to_ints(eng.commit(polys[0], 0))[0] == want[0]
    got = eng.commit_batch(polys, 0)
    for j in range(3):
        assert cops.affine_arr_to_ints(got[j:j + 1])[0] == want[j], (bits, j)
    # linearity (size-independent): commit(2 a) == 2 commit(a)
    twice = cops.fr_mont([2 * x % F.R for x in cops.fr_ints(cols[0])])
    p2 = eng.poly(n, twice)
    assert cops.affine_arr_to_ints(eng.commit(p2, 0))[0] == C.add(want[0], want[0])
    p2.free()
    for p in polys:
        p.free()


@pytest.mark.parametrize("bits", [0, 15])
def test_wide_commit_at_the_24_bit_index_boundary(windowed, bits):
    """k = 20: 16 windows x 2^20 points = 2^24 table entries fill the 24-bit index field of an entry exactly (the default
    plan there), 17 windows of 15 bits need the 25-bit layout."""
    eng = windowed(bits)
    k = 20
    n = 1 << k
    eng.srs_setup(k)
    assert eng.srs_msm_plan() == ((15, 17) if bits else (16, 16))
    a = rand_col(np.random.default_rng(20), n)
    a[n - 5:] = np.array([0xFFFFFFFFFFFFFFFF, 0xFFFFFFFFFFFFFFFF, 0xFFFFFFFFFFFFFFFF, 0x0FFFFFFFFFFFFFFF], dtype=np.uint64)  # the last points, every window
    p = eng.poly(n, a)
    assert cops.affine_arr_to_ints(eng.commit(p, 0))[0] == tau_commit(a)
    q = eng.poly(n, a)
    eng.lagrange_to_coeff(q)
    assert (eng.commit(p, 1) == eng.commit(q, 0)).all()
    p.free()
    q.free()


@pytest.mark.gpu
@pytest.mark.parametrize("m", [1, 1000, (1 << 16) - 3])
def test_wide_partial_length_msm_srs(windowed, m):
    """zk_msm_srs with fewer scalars than the SRS has points (ParamsKZG::commit of a short polynomial) on the wide path: the
    same point as the zero-padded full-length commitment (window tables strided by the SRS size, lanes and parts by m)."""
    from webauthn_halo2_amd import engine as E
    eng = windowed(16)
    k = 16
    n = 1 << k
    eng.srs_setup(k)
    a = rand_col(np.random.default_rng(77 + m), n)
    padded = a.copy()
    padded[m:] = 0
    want = eng.commit(eng.poly(n, padded), E.ZK_BASIS_MONOMIAL)
    got = cops.jac_to_affine_ints(eng.msm_srs(a[:m], E.ZK_BASIS_MONOMIAL))
    assert got == cops.affine_arr_to_ints(want.reshape(1, 8))[0]
    assert cops.affine_arr_to_ints(want.reshape(1, 8))[0] == tau_commit(padded)


@pytest.mark.parametrize("bits", [15, 16])
@pytest.mark.parametrize("k", [7, 9])
def test_window_override_below_the_workspace_floor(windowed, bits, k):
    """A 15 / 16-bit override on an SRS shorter than the 1024-scalar workspace floor (the table builder and the
    runner decided 'wide path' by different rules there): the override is clamped, commitments stay exact, batches work."""
    eng = windowed(bits)
    n = 1 << k
    eng.srs_setup(k)
    assert eng.srs_msm_plan()[0] <= 14
    rng = np.random.default_rng(77 * bits + k)
    cols = [rand_col(rng, n) for _ in range(3)]
    polys = [eng.poly(n, c) for c in cols]
    got = cops.affine_arr_to_ints(eng.commit_batch(polys, 0))
    for c, g, p in zip(cols, got, polys):
        assert g == tau_commit(c)
        assert cops.affine_arr_to_ints(eng.commit(p, 0))[0] == g
        p.free()


def test_wide_commit_with_25_bit_table_indexes(windowed):
    """k = 21 (BASELINE configs[4]): 16 windows x 2^21 points need 25 index bits per entry, which leaves 6 for the fine key
    (64 buckets per coarse bin, 512 bins) and 5 for the distance field — the wide path's second entry layout.  Random
    scalars, a column of 18-bit values (hot low buckets, most first-of-bucket entries escape) and the last points."""
    eng = windowed(0)
    k = 21
    n = 1 << k
    eng.srs_setup(k)
    assert eng.srs_msm_plan() == (16, 16)
    rng = np.random.default_rng(0x5EED0021)
    a = rand_col(rng, n)
    a[n - 3:] = np.array([0xFFFFFFFFFFFFFFFF, 0xFFFFFFFFFFFFFFFF, 0xFFFFFFFFFFFFFFFF, 0x0FFFFFFFFFFFFFFF], dtype=np.uint64)
    b = rand_col(rng, n)
    b[:, 1:] = 0
    b[:, 0] &= 0x3FFFF
    pa, pb = eng.poly(n, a), eng.poly(n, b)
    assert cops.affine_arr_to_ints(eng.commit(pa, 0))[0] == tau_commit(a)
    assert cops.affine_arr_to_ints(eng.commit(pb, 0))[0] == tau_commit(b)
    pa.free()
    pb.free()


@pytest.mark.parametrize("k", [13, 16])
def test_t1_per_bucket_and_per_part_give_the_same_commitments(windowed, k):
    """ZK_OPT_MSM_T1 (round 6): the reduction tail's first kernel as one lane per bucket (msm_wbucket_kernel; buckets with more
    than 24 partial sums stay with the part kernel) against parts of <= 8 + a segmented tree (msm_wparts_kernel) — the tau-oracle's
    commitments either way, for uniformly random columns (every bucket small), witness-like ones (a few giant buckets: both
    kernels of the per-bucket form take part), the all-in-one-bucket column and the zero column, alone and in one batched pass."""
    from webauthn_halo2_amd import engine as E

    eng = windowed(16)
    n = 1 << k
    eng.srs_setup(k)
    rng = np.random.default_rng(77 + k)
    pr = random.Random(k)
    mix = cops.fr_mont([pr.randrange(1 << 18) if pr.random() < 0.5 else pr.randrange(F.R) if pr.random() < 0.6 else 0 for _ in range(n)])
    boolc = cops.fr_mont([pr.randrange(2) for _ in range(n)])
    ones = cops.fr_mont([sum(1 << (16 * w) for w in range(15))] * n)
    data = [rand_col(rng, n), rand_col(rng, n), mix, boolc, ones, np.zeros((n, 4), dtype=np.uint64)]
    polys = [eng.poly(n, d) for d in data]
    want = [tau_commit(d) for d in data]
    try:
        for mode in (1, 2, 0):
            eng.set_option(E.ZK_OPT_MSM_T1, mode)
            for j, p in enumerate(polys):
                assert cops.affine_arr_to_ints(eng.commit(p, 0))[0] == want[j], (mode, k, j)
            got = eng.commit_batch(polys, 0)
            for j in range(len(polys)):
                assert cops.affine_arr_to_ints(got[j:j + 1])[0] == want[j], (mode, k, "batch", j)
    finally:
        eng.set_option(E.ZK_OPT_MSM_T1, 0)
        for p in polys:
            p.free()
