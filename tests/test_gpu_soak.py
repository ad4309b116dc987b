"""Determinism soak inside `-m gpu` (round 6; until round 5 it was tools/soak.py on the builder's lease only).

Why it is a test: round 5's first transform stream decided per CALL where a proof's column transforms run; under four pipelines the
count of active contexts dips to one now and then, a proof changed streams half-way, two streams' NTTs shared the context's
ping-pong scratch without an order between them — and `zk_prove` returned ZK_OK with wrong bytes for about 1 proof in 1 500.  None
of the parity tests saw it (their largest concurrent case is 32 jobs).  The reference's handlers `.unwrap()` the prover's result and
the proof goes on chain (proving-server/src/main.rs:60,76): a prover that is silently wrong once in a while is worse than a slow one.

What runs, in <= 120 s on one MI355X:
  * k = 17, EVM transcript + GWC (the proving server's compiled-in configuration, main.rs:17): four pipelines, 3 000 proofs, every
    stream option on auto, forced onto the side streams and forced onto the main stream in turn;
  * k = 19, Blake2b + SHPLONK (the headline configuration): four pipelines, 746 proofs, lone `zk_prove` calls alternating with
    lock-step batches of four (`zk_prove_batch`) under the same three stream regimes, plus a BURSTY auto phase — pipelines that pause
    between proofs, so that the number of active contexts keeps crossing the thresholds of the auto rules (1 <-> 2 <-> 4) in the
    middle of the others' proofs: the situation the round-5 race needed.
Every proof is compared with the bytes the first pipeline made for that job on its own before the soak (k = 17), and with the
ORACLE's committed digest of that job's proof (k = 19: tests/golden/batch_k19_sha256.json).

Checked once against the library of commit 28762db^ (the last one with the per-call decision), same box, same test file:
see the docstring of test_soak_k19_lone_and_lockstep for what it found.
"""
import hashlib
import json
import os
import threading
import time

import numpy as np
import pytest

import webauthn_halo2_amd as zk
from webauthn_halo2_amd import batch, circuit
from webauthn_halo2_amd import engine as E

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (tail stream, transform stream, MSM stream): 0 auto, 1 side stream, 2 main stream (include/zkmi355.h ZK_OPT_*)
REGIMES = {"auto": (0, 0, 0), "side": (1, 1, 1), "main": (2, 2, 2)}


def _pipelines(p, n):
    fixed, copies = batch.structure(p)
    pipes = [batch.Pipeline(0, p, fixed, copies, deterministic_seeds=True)]
    for _ in range(n - 1):
        pipes.append(batch.Pipeline(0, p, fixed, copies, deterministic_seeds=True, share_srs_with=pipes[0]))
    return pipes


def _set_regime(pipes, name):
    t, x, m = REGIMES[name]
    for pl in pipes:
        pl.eng.set_option(E.ZK_OPT_MSM_TAIL_STREAM, t)
        pl.eng.set_option(E.ZK_OPT_XFORM_STREAM, x)
        if hasattr(E, "ZK_OPT_MSM_STREAM"):  # (absent in the pre-fix library this file was also run against)
            pl.eng.set_option(E.ZK_OPT_MSM_STREAM, m)


def _threads(pipes, work):
    errs = []

    def guarded(q):
        try:
            work(q, pipes[q])
        except Exception as e:  # noqa: BLE001 — surfaced below
            errs.append(e)

    ths = [threading.Thread(target=guarded, args=(q,)) for q in range(len(pipes))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    if errs:
        raise errs[0]


def test_soak_k17_evm_four_pipelines():
    """3 000 k = 17 EVM + GWC proofs (2 720 bytes, the /prove_evm configuration) over four pipelines, 1 000 per stream regime."""
    t_start = time.time()
    p = circuit.K17
    jobs = list(range(4))
    wit = batch.synthesize_jobs(p, jobs)
    pipes = _pipelines(p, 4)
    try:
        for pl in pipes:
            for j in jobs:
                pl.load(j, wit[j])
        ref = {j: pipes[0].prove(j, E.ZK_TRANSCRIPT_EVM, keep=True) for j in jobs}
        assert all(len(v) == 2720 for v in ref.values()) and len(set(ref.values())) == len(jobs)
        # the other pipelines' keys and workspaces, each on its own: same bytes
        for pl in pipes[1:]:
            assert pl.prove(jobs[1], E.ZK_TRANSCRIPT_EVM, keep=True) == ref[jobs[1]]
        bad, done = [], [0]
        per = 250
        for regime in REGIMES:
            _set_regime(pipes, regime)

            def work(q, pl):
                for i in range(per):
                    j = jobs[(i + q) % len(jobs)]
                    if pl.prove(j, E.ZK_TRANSCRIPT_EVM, keep=True) != ref[j]:
                        bad.append((regime, q, i, j))
                done[0] += per

            _threads(pipes, work)
        assert done[0] >= 3000
        assert not bad, "k = 17 EVM proofs that changed their bytes under concurrency: %s" % bad[:8]
    finally:
        for pl in pipes[::-1]:
            pl.close()
    print("soak k=17: %d proofs in %.1f s (set-up included)" % (done[0], time.time() - t_start))


def test_soak_k19_lone_and_lockstep():
    """746 k = 19 Blake2b + SHPLONK proofs over four pipelines against the oracle's committed digests: lone proofs and lock-step
    batches of four alternating, stream options auto / side / main, and a bursty auto phase whose pipelines pause between proofs so
    that the active-context count crosses the auto rules' thresholds while other proofs are half-way, and a BLINKER phase: one prover
    beside a context that commits every few milliseconds, so that the count flips several times inside every proof.

    Against the library of 28762db^ (per-call transform-stream decision), one box, this file: recorded in
    profiles/r6_soak_on_prefix_commit.txt."""
    t_start = time.time()
    p = circuit.K19
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "batch_k19_sha256.json")))["sha256"]
    jobs = list(range(8))
    wit = batch.synthesize_jobs(p, jobs)
    pipes = _pipelines(p, 4)
    bad, count = [], [0]
    lock = threading.Lock()

    def check(tag, q, j, proof):
        with lock:
            count[0] += 1
            if len(proof) != 960 or hashlib.sha256(proof).hexdigest() != want[str(j)]:
                bad.append((tag, q, j))

    try:
        for pl in pipes:
            for j in jobs:
                pl.load(j, wit[j])
        for j in jobs[:2]:
            check("alone", 0, j, pipes[0].prove(j, E.ZK_TRANSCRIPT_BLAKE2B, keep=True))
        for regime in REGIMES:
            _set_regime(pipes, regime)

            def work(q, pl):
                for r in range(5):  # 5 x (a batch of four + four lone proofs) = 40 proofs per pipeline and regime
                    group = [jobs[(4 * r + q + t) % len(jobs)] for t in range(4)]
                    for j, pf in zip(group, pl.prove_lockstep(group, E.ZK_TRANSCRIPT_BLAKE2B, keep=True)):
                        check(regime + "/lockstep", q, j, pf)
                    for j in group:
                        check(regime + "/lone", q, j, pl.prove(j, E.ZK_TRANSCRIPT_BLAKE2B, keep=True))

            _threads(pipes, work)
        # bursty: auto rules, lone proofs, pipelines that stop and start out of step with each other
        _set_regime(pipes, "auto")
        rng = np.random.default_rng(0x50AC)
        pauses = rng.integers(0, 4, size=(4, 36))

        def bursty(q, pl):
            for i in range(36):
                j = jobs[(i + 3 * q) % len(jobs)]
                check("bursty", q, j, pl.prove(j, E.ZK_TRANSCRIPT_BLAKE2B, keep=True))
                if q and pauses[q][i]:
                    time.sleep(0.004 * int(pauses[q][i]) * q)  # 4 .. 36 ms: longer than the 4 ms activity window

        _threads(pipes, bursty)
        # blinker: ONE pipeline proves lone proofs while a second context of the device makes a short commitment every 5 .. 9 ms.
        # Every MSM pass stamps its context in the device's activity table and a stamp counts for 4 ms, so the prover sees the number
        # of active contexts go 2 -> 1 -> 2 several times inside each 11 ms proof: every auto rule that is read more than once per
        # proof (round 5's first transform-stream rule was) flips between two of its reads
        blink = zk.Engine(0, share_with=pipes[0].eng)
        short = blink.poly(1 << p.degree, np.ones((1 << p.degree, 4), dtype=np.uint64))
        stop = threading.Event()

        def blinker():
            r = np.random.default_rng(7)
            while not stop.is_set():
                blink.commit(short, 1)
                time.sleep(0.005 + 0.004 * float(r.random()))

        bt = threading.Thread(target=blinker)
        bt.start()
        try:
            for i in range(120):
                j = jobs[i % len(jobs)]
                check("blinker", 0, j, pipes[0].prove(j, E.ZK_TRANSCRIPT_BLAKE2B, keep=True))
        finally:
            stop.set()
            bt.join()
            short.free()
            blink.close()
        assert count[0] >= 600
        assert not bad, "k = 19 proofs that differ from the oracle's digests: %s" % bad[:8]
    finally:
        for pl in pipes[::-1]:
            pl.close()
    print("soak k=19: %d proofs in %.1f s (set-up included)" % (count[0], time.time() - t_start))
