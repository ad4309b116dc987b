"""ZK_OPT_STREAM_AUDIT (csrc/audit.h): the happens-before ledger of a context's streams.

Round 5's soak found a cross-stream race (two streams' NTTs on one scratch) that made `zk_prove` return ZK_OK with wrong bytes about
once in 1 500 proofs; the reference's handlers `.unwrap()` whatever the prover returns and the proof goes on chain
(proving-server/src/main.rs:60,76).  The audit checks the ORDER the engine establishes between its streams, not the timing: with
the option on every enqueue names the buffers it reads and writes, and one that is not ordered after a buffer's last writer (or,
for a write, its last readers) turns the entry point's ZK_OK into ZK_EINTERNAL.  Here:
  * the prover tests' shapes run ONCE MORE under the audit, in every stream regime the engine has (auto, everything forced onto the
    side streams, everything on the main stream), lone, two contexts side by side, and as lock-step batches: the oracle's bytes,
    no violation, and the ledger did check something;
  * the audit's SELF-TEST (option value 2) makes the prover take round 5's faulty path on purpose (the transform stream chosen per
    call, no order on the shared scratch): every such proof must be refused, and the report must name the hazard."""
import threading

import numpy as np
import pytest

import webauthn_halo2_amd as zk
from webauthn_halo2_amd import engine as E
from zkoracle import plonk, prover
from zkoracle.hashes import ChaCha20Rng

pytestmark = pytest.mark.gpu

SHAPES = {
    "k10single": (1, 1, 1, 10, 9, 0),     # the k = 19 column shape on the window tables (pipelined advice pass)
    "k10batched": (3, 2, 1, 10, 8, 0),    # several columns per MSM pass, two lookups, three lanes in flight
    "k17like": (4, 1, 1, 7, 5, 0),
    "idle": (5, 2, 2, 7, 5, 2),
    "manycols_k8": (44, 6, 2, 8, 6, 0),   # argument blocks in device memory, staging reused behind synchronisations
}
REGIMES = {"auto": (0, 0, 0), "side": (1, 1, 1), "main": (2, 2, 2)}
KIND = {"evm": E.ZK_TRANSCRIPT_EVM, "blake2b": E.ZK_TRANSCRIPT_BLAKE2B}


def _make(eng, shape, seeds):
    A, L, F, k, lb, idle = shape
    p = zk.circuit.CircuitParams(degree=k, num_advice=A, num_lookup_advice=L, num_fixed=F, lookup_bits=lb, idle_gate_columns=idle)
    asgs = [zk.circuit.synthesize(p, s) for s in seeds]
    fixed = np.stack([asgs[0].to_limbs(c) for c in asgs[0].fixed])
    pk = eng.keygen(p, fixed, asgs[0].copies)
    sets = []
    for asg in asgs:
        hs = []
        for col in asg.advice:
            h = eng.poly(1 << k)
            eng.upload_canonical(h, asg.to_limbs(col))
            hs.append(h)
        sets.append(hs)
    opk = prover.keygen(prover.Circuit(plonk.Shape(k, A, L, F, lb, idle), asgs[0].fixed, asgs[0].copies, asgs[0].advice))
    return pk, sets, asgs, opk


def _regime(eng, name):
    t, x, m = REGIMES[name]
    eng.set_option(E.ZK_OPT_MSM_TAIL_STREAM, t)
    eng.set_option(E.ZK_OPT_XFORM_STREAM, x)
    eng.set_option(E.ZK_OPT_MSM_STREAM, m)


@pytest.mark.parametrize("name", list(SHAPES))
def test_prover_shapes_under_the_audit(name):
    eng = zk.Engine(0)
    eng.set_option(E.ZK_OPT_STREAM_AUDIT, 1)
    k = SHAPES[name][3]
    eng.srs_setup(k)
    seeds = [0x5EED0019, 0x5EED0019 + 11, 0x5EED0019 + 23]
    pk, sets, asgs, opk = _make(eng, SHAPES[name], seeds)
    rs = [bytes([31 + i]) * 32 for i in range(3)]
    for regime in REGIMES:
        _regime(eng, regime)
        for kind in ("blake2b", "evm"):
            for j in range(2):
                try:
                    got1 = eng.prove(pk, sets[j], rs[j], KIND[kind])
                except zk.ZkError as e:
                    raise AssertionError("%s under the audit (%s, %s): %s" % (e, name, regime, eng.audit_report())) from e
                assert got1 == prover.create_proof(opk, asgs[j].advice, ChaCha20Rng(rs[j]), kind), (name, regime, kind, j)
            try:
                got = eng.prove_batch(pk, sets, rs, KIND[kind])  # lock-step: members' workspaces, wider passes
            except zk.ZkError as e:
                raise AssertionError("%s (batch) under the audit (%s, %s): %s" % (e, name, regime, eng.audit_report())) from e
            assert got == [prover.create_proof(opk, asgs[j].advice, ChaCha20Rng(rs[j]), kind) for j in range(3)], (name, regime, kind)
        assert eng.prove(pk, sets[0], rs[0], E.ZK_TRANSCRIPT_EVM, E.ZK_SCHEME_SHPLONK) == prover.create_proof(opk, asgs[0].advice, ChaCha20Rng(rs[0]), "evm", "shplonk")
        assert eng.prove(pk, sets[0], rs[0], E.ZK_TRANSCRIPT_BLAKE2B, E.ZK_SCHEME_GWC) == prover.create_proof(opk, asgs[0].advice, ChaCha20Rng(rs[0]), "blake2b", "gwc")
    checks, violations, msg = eng.audit_report()
    assert violations == 0, msg
    assert checks > 1000, checks  # the ledger saw the proofs' enqueues
    eng.close()


def test_two_contexts_side_by_side_under_the_audit():
    """Two pipelines of one device (shared SRS), each with its own ledger, proving concurrently under the auto rules: the number of
    active contexts — what the auto rules read — changes while proofs are half-way."""
    engs = [zk.Engine(0)]
    engs[0].set_option(E.ZK_OPT_STREAM_AUDIT, 1)
    engs[0].srs_setup(10)
    engs.append(zk.Engine(0, share_with=engs[0]))
    engs[1].set_option(E.ZK_OPT_STREAM_AUDIT, 1)
    made = [_make(e, SHAPES["k10single"], [0x5EED0019]) for e in engs]
    seed = b"\x55" * 32
    want = {kind: prover.create_proof(made[0][3], made[0][2][0].advice, ChaCha20Rng(seed), kind) for kind in KIND}
    bad = []

    def work(i):
        pk, sets, _, _ = made[i]
        for r in range(30):
            kind = "evm" if r & 1 else "blake2b"
            try:
                if engs[i].prove(pk, sets[0], seed, KIND[kind]) != want[kind]:
                    bad.append((i, r, "bytes"))
            except zk.ZkError as e:
                bad.append((i, r, e.code, engs[i].audit_report()[2]))

    ths = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not bad, bad[:4]
    for e in engs:
        checks, violations, msg = e.audit_report()
        assert violations == 0 and checks > 1000, (checks, violations, msg)
    for e in engs[::-1]:
        e.close()


@pytest.mark.parametrize("name", ["k10single", "k10batched"])
def test_audit_self_test_refuses_the_unordered_path(name):
    """Option value 2: audit on AND the prover decides the transform stream per call (alternating) without ordering the shared NTT
    scratch — round 5's first form.  Every proof must come back as ZK_EINTERNAL (-7), the report must describe a hazard on the
    scratch between two NTT batches; switched back to 1 the same context proves the oracle's bytes again."""
    eng = zk.Engine(0)
    eng.srs_setup(SHAPES[name][3])
    pk, sets, asgs, opk = _make(eng, SHAPES[name], [0x5EED0019])
    seed = b"\x66" * 32
    want = prover.create_proof(opk, asgs[0].advice, ChaCha20Rng(seed), "blake2b")
    assert eng.prove(pk, sets[0], seed, E.ZK_TRANSCRIPT_BLAKE2B) == want
    eng.set_option(E.ZK_OPT_STREAM_AUDIT, 2)
    for _ in range(3):
        with pytest.raises(zk.ZkError) as e:
            eng.prove(pk, sets[0], seed, E.ZK_TRANSCRIPT_BLAKE2B)
        assert e.value.code == -7
    checks, violations, msg = eng.audit_report()
    assert violations >= 3 and "NTT batch" in msg and "hazard" in msg, (violations, msg)
    eng.set_option(E.ZK_OPT_STREAM_AUDIT, 1)
    assert eng.prove(pk, sets[0], seed, E.ZK_TRANSCRIPT_BLAKE2B) == want
    assert eng.audit_report()[1] == 0
    eng.close()


def test_lone_k19_proof_on_four_streams_under_the_audit():
    """BASELINE's size under the AUTO rules: a lone k = 19 proof is the one case that really runs on four streams (main + tail +
    transform + MSM stream: the rules need k >= 18 and a quiet device) — two proofs of the batch workload's jobs under the audit,
    their digests the oracle's, no violation; then the same with the transforms back on the main stream."""
    import hashlib
    import json
    import os

    from webauthn_halo2_amd import batch, circuit

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    want = json.load(open(os.path.join(root, "tests", "golden", "batch_k19_sha256.json")))["sha256"]
    p = circuit.K19
    wit = batch.synthesize_jobs(p, [0, 1])
    pl = batch.Pipeline(0, p, deterministic_seeds=True)
    try:
        pl.eng.set_option(E.ZK_OPT_STREAM_AUDIT, 1)
        for j in (0, 1):
            pl.load(j, wit[j])
        for regime in ("auto", "main", "auto"):
            _regime(pl.eng, regime)
            for j in (0, 1):
                try:
                    pf = pl.prove(j, E.ZK_TRANSCRIPT_BLAKE2B, keep=True)
                except zk.ZkError as e:
                    raise AssertionError("%s under the audit (k = 19, %s): %s" % (e, regime, pl.eng.audit_report())) from e
                assert hashlib.sha256(pf).hexdigest() == want[str(j)], (regime, j)
        checks, violations, msg = pl.eng.audit_report()
        assert violations == 0 and checks > 500, (checks, violations, msg)
    finally:
        pl.close()
