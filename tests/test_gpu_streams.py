"""Where the MSM reduction tails run (ZK_OPT_MSM_TAIL_STREAM auto mode) must depend on how busy the DEVICE is, not on which
entry point the host uses: round 4 counted zk_prove calls only, so four threads on the phase-level ABI
(the Rust shim of INTEGRATION.md) stayed in the slow regime."""
import threading
import time

import numpy as np
import pytest

import webauthn_halo2_amd as zk
from webauthn_halo2_amd import engine as E

pytestmark = pytest.mark.gpu


def _column(n, seed):
    a = np.frombuffer(np.random.default_rng(seed).bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
    a[:, 3] &= 0x0FFFFFFFFFFFFFFF
    return a


def test_tail_placement_follows_device_activity_whatever_the_entry_point():
    k = 14
    n = 1 << k
    engs = [zk.Engine(0)]
    engs[0].srs_setup(k)
    for _ in range(3):
        engs.append(zk.Engine(0, share_with=engs[0]))
    polys = [e.poly(n, _column(n, 7 + i)) for i, e in enumerate(engs)]
    want = [e.commit(p, E.ZK_BASIS_LAGRANGE).copy() for e, p in zip(engs, polys)]
    time.sleep(0.02)  # the set-up commits above fall out of the activity window
    # (a) one context alone: every tail on its side stream
    engs[0].timer_reset()
    for _ in range(20):
        engs[0].commit(polys[0], E.ZK_BASIS_LAGRANGE)
    assert engs[0].timer_stats(E.ZK_T_MSM_TAIL_MAIN)[1] == 0
    assert engs[0].timer_stats(E.ZK_T_MSM)[1] == 20
    # (b) four host threads, each on zk_commit of its own context: the device is as busy as under four zk_prove calls
    time.sleep(0.02)
    for e in engs:
        e.timer_reset()
    reps = 300
    errs = []
    go = threading.Barrier(4)

    def work(i):
        try:
            go.wait()
            for _ in range(reps):
                got = engs[i].commit(polys[i], E.ZK_BASIS_LAGRANGE)
            assert np.array_equal(got, want[i])
        except Exception as ex:  # surfaced below
            errs.append(ex)

    ths = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    main = sum(e.timer_stats(E.ZK_T_MSM_TAIL_MAIN)[1] for e in engs)
    total = sum(e.timer_stats(E.ZK_T_MSM)[1] for e in engs)
    assert total == 4 * reps
    assert main >= 0.8 * total, (main, total)  # the start and the end of the run see fewer than three active contexts
    # (c) the threshold is an option: above 8 active contexts never happens here -> side stream again
    time.sleep(0.02)
    for e in engs:
        e.set_option(E.ZK_OPT_MSM_TAIL_MAIN_ABOVE, 8)
        e.timer_reset()
    ths = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    go.reset()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    assert sum(e.timer_stats(E.ZK_T_MSM_TAIL_MAIN)[1] for e in engs) == 0
    # (d) pinned: main stream for a lone caller
    engs[0].set_option(E.ZK_OPT_MSM_TAIL_STREAM, 2)
    engs[0].timer_reset()
    assert np.array_equal(engs[0].commit(polys[0], E.ZK_BASIS_LAGRANGE), want[0])
    assert engs[0].timer_stats(E.ZK_T_MSM_TAIL_MAIN)[1] == 1
    for p in polys:
        p.free()
    for e in engs[::-1]:
        e.close()


@pytest.mark.parametrize("mode", [1, 2], ids=["transforms-on-their-own-stream", "transforms-on-the-main-stream"])
def test_transform_stream_placement_does_not_change_proofs(mode):
    """ZK_OPT_XFORM_STREAM / ZK_OPT_MSM_STREAM pin where a proof's column transforms and MSM passes run (auto: streams of their
    own for a lone k >= 18 proof, decided once per proof): the oracle's bytes either way, for a lone proof and for two pipelines proving side by side, at a
    multi-column and a one-column shape whose commitments run on the window tables."""
    import threading as th

    from zkoracle import plonk, prover
    from zkoracle.hashes import ChaCha20Rng

    for (A, L, F, k, lb) in ((3, 2, 1, 10, 8), (1, 1, 1, 10, 9)):
        p = zk.circuit.CircuitParams(degree=k, num_advice=A, num_lookup_advice=L, num_fixed=F, lookup_bits=lb)
        asg = zk.circuit.synthesize(p, 0x5EED0019)
        sh = plonk.Shape(k, A, L, F, lb)
        opk = prover.keygen(prover.Circuit(sh, asg.fixed, asg.copies, asg.advice))
        fixed = np.stack([asg.to_limbs(c) for c in asg.fixed])
        engs = [zk.Engine(0)]
        engs[0].set_option(E.ZK_OPT_XFORM_STREAM, mode)
        engs[0].set_option(E.ZK_OPT_MSM_STREAM, mode)  # the MSM passes on their own stream too (or both on the main stream)
        engs[0].srs_setup(k)
        engs.append(zk.Engine(0, share_with=engs[0]))
        pks, cols = [], []
        for e in engs:
            pks.append(e.keygen(p, fixed, asg.copies))
            hs = []
            for col in asg.advice:
                h = e.poly(1 << k)
                e.upload_canonical(h, asg.to_limbs(col))
                hs.append(h)
            cols.append(hs)
        seed = bytes([40 + mode]) * 32
        want = {kind: prover.create_proof(opk, asg.advice, ChaCha20Rng(seed), kind) for kind in ("blake2b", "evm")}
        tk = {"blake2b": E.ZK_TRANSCRIPT_BLAKE2B, "evm": E.ZK_TRANSCRIPT_EVM}
        for kind in want:
            assert engs[0].prove(pks[0], cols[0], seed, tk[kind]) == want[kind]
        bad = []

        def work(i):
            for r in range(40):
                kind = "evm" if r & 1 else "blake2b"
                if engs[i].prove(pks[i], cols[i], seed, tk[kind]) != want[kind]:
                    bad.append((i, r))

        ths = [th.Thread(target=work, args=(i,)) for i in range(2)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        assert not bad, bad
        for e in engs[::-1]:
            e.close()


def test_stream_priority_option_leaves_results_unchanged():
    """ZK_OPT_STREAM_PRIORITY (experiment, profiles/r6_ab_pipeline_priority.txt) re-makes a context's main stream at another
    dispatch priority: same commitments, bad values refused, and a context that shares the first one's SRS takes it too."""
    k = 12
    n = 1 << k
    base = zk.Engine(0)
    base.srs_setup(k)
    col = _column(n, 99)
    want = base.commit(base.poly(n, col), E.ZK_BASIS_LAGRANGE).copy()
    for value in (1, 2, 0):
        e = zk.Engine(0, share_with=base)
        e.set_option(E.ZK_OPT_STREAM_PRIORITY, value)
        assert np.array_equal(e.commit(e.poly(n, col), E.ZK_BASIS_LAGRANGE), want)
        e.close()
    with pytest.raises(zk.ZkError):
        base.set_option(E.ZK_OPT_STREAM_PRIORITY, 3)
    base.set_option(E.ZK_OPT_STREAM_PRIORITY, 1)  # on a context that has already worked: drained, then replaced
    assert np.array_equal(base.commit(base.poly(n, col), E.ZK_BASIS_LAGRANGE), want)


def test_stream_pool_more_contexts_than_slots_and_reuse():
    """The per-device stream pool (engine.hip): eight slots of streams made in a fixed order, taken by zk_ctx_create and given back
    by zk_ctx_destroy; contexts beyond the slots make their own streams.  Twelve contexts at once, then again after all were
    destroyed: every context commits the same column to the same point, alone and from twelve threads."""
    k = 12
    n = 1 << k
    col = _column(n, 5)
    want = None
    for round_ in range(2):
        engs = [zk.Engine(0)]
        engs[0].srs_setup(k)
        for _ in range(11):
            engs.append(zk.Engine(0, share_with=engs[0]))
        polys = [e.poly(n, col) for e in engs]
        got = [e.commit(p, E.ZK_BASIS_LAGRANGE).copy() for e, p in zip(engs, polys)]
        want = got[0] if want is None else want
        assert all(np.array_equal(g, want) for g in got)
        out = [None] * len(engs)

        def work(i):
            for _ in range(30):
                out[i] = engs[i].commit(polys[i], E.ZK_BASIS_LAGRANGE).copy()

        ths = [threading.Thread(target=work, args=(i,)) for i in range(len(engs))]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        assert all(np.array_equal(o, want) for o in out)
        for e in engs[::-1]:
            e.close()


def test_activity_hold_option_same_bytes():
    """ZK_OPT_ACTIVITY_HOLD: a context inside zk_prove counts as active for the whole call (default) or by its stamps alone (1) —
    placement only: two pipelines proving side by side give the same bytes under either rule."""
    from webauthn_halo2_amd import batch, circuit

    p = circuit.CircuitParams(degree=10, num_advice=3, num_lookup_advice=2, num_fixed=1, lookup_bits=8)
    fixed, copies = batch.structure(p)
    wit = batch.synthesize_jobs(p, [0, 1, 2, 3], processes=1)
    proofs = {}
    for rule in (0, 1):
        def factory(dev, rule=rule):
            e = zk.Engine(dev)
            e.set_option(E.ZK_OPT_ACTIVITY_HOLD, rule)
            return e

        factory.configure = lambda e, rule=rule: e.set_option(E.ZK_OPT_ACTIVITY_HOLD, rule)
        pipes = [batch.Pipeline(0, p, fixed, copies, engine_factory=factory, deterministic_seeds=True)]
        pipes.append(batch.Pipeline(0, p, fixed, copies, engine_factory=factory, deterministic_seeds=True, share_srs_with=pipes[0]))
        for q, pl in enumerate(pipes):
            for j in (0, 1, 2, 3)[q::2]:
                pl.load(j, wit[j])
        proofs[rule] = batch.run(pipes, [0, 1, 2, 3])
        for pl in pipes[::-1]:
            pl.close()
    assert proofs[0] == proofs[1] and len(set(proofs[0].values())) == 4
    with pytest.raises(zk.ZkError):
        zk.Engine(0).set_option(E.ZK_OPT_ACTIVITY_HOLD, 3)
    # 2: a context held active by its host (phase-level ABI): three such contexts, idle, and a fourth one's lone commits take the
    # loaded regime (tails on the main stream); released again, they go back to the side stream
    k = 12
    n = 1 << k
    engs = [zk.Engine(0)]
    engs[0].srs_setup(k)
    for _ in range(3):
        engs.append(zk.Engine(0, share_with=engs[0]))
    poly = engs[3].poly(n, _column(n, 3))
    engs[3].commit(poly, E.ZK_BASIS_LAGRANGE)
    time.sleep(0.02)
    for value, on_main in ((2, 10), (0, 0)):
        for e in engs[:3]:
            e.set_option(E.ZK_OPT_ACTIVITY_HOLD, value)
        time.sleep(0.02)  # (a context that lets go counts as just-active for the 4 ms window)
        engs[3].timer_reset()
        for _ in range(10):
            engs[3].commit(poly, E.ZK_BASIS_LAGRANGE)
        assert engs[3].timer_stats(E.ZK_T_MSM_TAIL_MAIN)[1] == on_main, value
    for e in engs[::-1]:
        e.close()
