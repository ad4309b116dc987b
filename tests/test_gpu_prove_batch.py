"""zk_prove_batch (lock-step proving of B independent proofs on one context, csrc/prover_batch.h): every proof of a batch is
byte-identical to the oracle's create_proof — and therefore to zk_prove — for the same witness and ChaCha20 stream, for every
column shape the engine branches on, both transcripts / multi-open schemes, batch sizes that fill a pass, overflow it, and
leave a remainder; at BASELINE's size the batch's proofs equal the committed digests of configs[3]."""
import hashlib
import json
import os

import numpy as np
import pytest

import webauthn_halo2_amd as zk
from webauthn_halo2_amd import engine as E
from zkoracle import plonk, prover
from zkoracle.hashes import ChaCha20Rng

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
KIND = {"evm": E.ZK_TRANSCRIPT_EVM, "blake2b": E.ZK_TRANSCRIPT_BLAKE2B}
# (A, L, F, k, lookup_bits, idle): the shapes of tests/test_gpu_prover.py that take different branches of the prover
SHAPES = {
    "k19like": (1, 1, 1, 7, 6, 0),        # one advice column, one lookup: the pipelined advice pass
    "k17like": (4, 1, 1, 7, 5, 0),        # several gate columns, three permutation chunks: chained grand products per proof
    "wide": (3, 2, 2, 8, 6, 0),           # two lookups, two constants columns
    "idle": (5, 2, 2, 7, 5, 2),           # combined selectors
    "k10batched": (3, 2, 1, 10, 8, 0),    # window tables exist: column-batched MSM passes on the wide path's little brother
    "k10single": (1, 1, 1, 10, 9, 0),
    "manycols": (36, 12, 2, 7, 5, 0),     # argument blocks in device memory (more than 8 chunks / lookups / columns)
}


def _setup(eng, shape, seeds):
    A, L, F, k, lb, idle = shape
    p = zk.circuit.CircuitParams(degree=k, num_advice=A, num_lookup_advice=L, num_fixed=F, lookup_bits=lb, idle_gate_columns=idle)
    asgs = [zk.circuit.synthesize(p, s) for s in seeds]  # one structure (STRUCT_SEED), a witness per seed
    eng.srs_setup(k)
    fixed = np.stack([asgs[0].to_limbs(c) for c in asgs[0].fixed])
    pk = eng.keygen(p, fixed, asgs[0].copies)
    sets = []
    for asg in asgs:
        polys = []
        for col in asg.advice:
            h = eng.poly(1 << k)
            eng.upload_canonical(h, asg.to_limbs(col))
            polys.append(h)
        sets.append(polys)
    sh = plonk.Shape(k, A, L, F, lb, idle)
    opk = prover.keygen(prover.Circuit(sh, asgs[0].fixed, asgs[0].copies, asgs[0].advice))
    return pk, sets, asgs, opk


@pytest.mark.parametrize("name", list(SHAPES))
def test_batch_proofs_byte_identical_to_oracle(name):
    eng = zk.Engine(0)
    B = 3 if name == "manycols" else 5
    seeds = [0x5EED0019 + 7 * i for i in range(B)]
    pk, sets, asgs, opk = _setup(eng, SHAPES[name], seeds)
    rng_seeds = [bytes([17 + i]) * 32 for i in range(B)]
    for kind in ("blake2b", "evm"):
        got = eng.prove_batch(pk, sets, rng_seeds, KIND[kind])
        assert len(got) == B
        for j in range(B):
            assert got[j] == prover.create_proof(opk, asgs[j].advice, ChaCha20Rng(rng_seeds[j]), kind), (name, kind, j)
            assert plonk.verify(opk.vk, got[j], kind)
    # both multi-open schemes under both transcripts, against the single prover (whose bytes the oracle pins elsewhere)
    for tr, scheme in ((E.ZK_TRANSCRIPT_BLAKE2B, E.ZK_SCHEME_GWC), (E.ZK_TRANSCRIPT_EVM, E.ZK_SCHEME_SHPLONK)):
        got = eng.prove_batch(pk, sets[:2], rng_seeds[:2], tr, scheme)
        for j in range(2):
            assert got[j] == eng.prove(pk, sets[j], rng_seeds[j], tr, scheme)
    # a batch of one is zk_prove; a smaller batch after a larger one reuses the workspaces
    assert eng.prove_batch(pk, sets[:1], rng_seeds[:1]) == [eng.prove(pk, sets[0], rng_seeds[0])]
    assert eng.prove_batch(pk, sets[1:3], rng_seeds[1:3]) == [eng.prove(pk, sets[j], rng_seeds[j]) for j in (1, 2)]
    for polys in sets:
        for h in polys:
            h.free()
    eng.pk_free(pk)
    eng.close()


@pytest.mark.parametrize("shape", ["k10batched", "k10single"])
@pytest.mark.parametrize("cols", [1, 3, 6, 16])
def test_pass_width_does_not_change_proofs(cols, shape):
    """ZK_OPT_BATCH_PASS_COLUMNS: one column per pass (every commitment its own pass), a width that splits a proof's columns
    across passes, one that makes the lookup commitments of a pipelined batch (one advice column, one lookup: the advice pass
    stays in flight on its own lane) need MORE passes than the two other lanes hold — the queue must then collect the advice
    pass and squeeze theta before it writes any a' —, and one wider than the default: the bytes stay the oracle's (k = 10: the
    MSM passes run on the window tables)."""
    eng = zk.Engine(0)
    eng.set_option(E.ZK_OPT_BATCH_PASS_COLUMNS, cols)
    B = 6
    seeds = [0x5EED0100 + i for i in range(B)]
    pk, sets, asgs, opk = _setup(eng, SHAPES[shape], seeds)
    rng_seeds = [bytes([3 * i + 1]) * 32 for i in range(B)]
    for kind in ("blake2b", "evm"):
        got = eng.prove_batch(pk, sets, rng_seeds, KIND[kind])
        for j in range(B):
            assert got[j] == prover.create_proof(opk, asgs[j].advice, ChaCha20Rng(rng_seeds[j]), kind), (cols, kind, j)
    eng.close()


def test_more_lookups_than_one_permutation_launch_takes():
    """Five proofs x twelve lookups = 60 lookups: more than one argument block of the lookup-permutation launches holds (56), so
    the batch's lookups go through two groups of launches on consecutive slices of the scratch; 5 x 37 grand products in one scan."""
    eng = zk.Engine(0)
    B = 5
    pk, sets, asgs, opk = _setup(eng, SHAPES["manycols"], [0x5EED0200 + i for i in range(B)])
    rng_seeds = [bytes([90 + i]) * 32 for i in range(B)]
    for kind in ("blake2b", "evm"):
        got = eng.prove_batch(pk, sets, rng_seeds, KIND[kind])
        for j in range(B):
            assert got[j] == prover.create_proof(opk, asgs[j].advice, ChaCha20Rng(rng_seeds[j]), kind), (kind, j)
    with pytest.raises(zk.ZkError) as e:  # 7 x 37 grand products: beyond the one-workgroup chain scan
        eng.prove_batch(pk, sets + sets[:2], rng_seeds + rng_seeds[:2])
    assert e.value.code == -1
    eng.close()


def test_batch_rejects_a_bad_witness_as_a_whole_and_recovers():
    """One proof's lookup input is off the table: the batch fails with ZK_EWITNESS (halo2: ConstraintSystemFailure), nothing
    is left in flight, and the same context proves the good jobs afterwards."""
    eng = zk.Engine(0)
    B = 3
    pk, sets, asgs, opk = _setup(eng, SHAPES["k19like"], [11, 12, 13])
    bad = [list(c) for c in asgs[1].advice]
    lay = asgs[1].layout
    ql = asgs[1].fixed[lay.fx_qlookup]
    row = next(r for r in range(lay.usable_rows) if ql[r])
    bad[0][row] = 1 << 20  # a looked-up cell outside 0 .. 2^6 - 1
    h = eng.poly(1 << 7)
    eng.upload_canonical(h, zk.circuit.Assignment.to_limbs(bad[0]))
    rng_seeds = [bytes([9 + i]) * 32 for i in range(B)]
    with pytest.raises(zk.ZkError) as e:
        eng.prove_batch(pk, [sets[0], [h], sets[2]], rng_seeds)
    assert e.value.code == -6
    got = eng.prove_batch(pk, sets, rng_seeds)
    for j in range(B):
        assert got[j] == prover.create_proof(opk, asgs[j].advice, ChaCha20Rng(rng_seeds[j]), "blake2b")
    with pytest.raises(zk.ZkError):
        eng.prove_batch(pk, [sets[0]] * 65, [bytes(32)] * 65)  # beyond ZK_PROVE_BATCH_MAX
    eng.close()


def test_k19_lockstep_batches_equal_committed_oracle_digests():
    """BASELINE configs[3] at its size through the lock-step prover: jobs 8 .. 23 of the 256-job batch as two pipelines x
    batches of four (bench.py's regime), and jobs 24 .. 29 as one batch of six (a remainder: three passes hold unequal
    shares), against the oracle's committed digests (tests/golden/batch_k19_sha256.json); the first job's full bytes against
    tests/golden/batch_k19_proofs.json."""
    from webauthn_halo2_amd import batch

    want = json.load(open(os.path.join(HERE, "golden", "batch_k19_sha256.json")))["sha256"]
    full = json.load(open(os.path.join(HERE, "golden", "batch_k19_proofs.json")))["jobs"]
    p = zk.circuit.K19
    fixed, copies = batch.structure(p)
    jobs = list(range(0, 24))
    wit = batch.synthesize_jobs(p, jobs + list(range(24, 30)))
    pipes = [batch.Pipeline(0, p, fixed, copies, deterministic_seeds=True)]
    pipes.append(batch.Pipeline(0, p, fixed, copies, deterministic_seeds=True, share_srs_with=pipes[0]))
    for q, pl in enumerate(pipes):
        for j in jobs[q::2]:
            pl.load(j, wit[j])
    got = batch.run_lockstep(pipes, jobs, 4, E.ZK_TRANSCRIPT_BLAKE2B)
    assert sorted(got) == jobs
    for j in jobs:
        assert len(got[j]) == 960 and hashlib.sha256(got[j]).hexdigest() == want[str(j)], j
    assert got[0].hex() == full["0"]["proof"]
    rest = list(range(24, 30))
    for j in rest:
        pipes[0].load(j, wit[j])
    six = pipes[0].prove_lockstep(rest)
    for j, pf in zip(rest, six):
        assert hashlib.sha256(pf).hexdigest() == want[str(j)], j
    for pl in pipes[::-1]:
        pl.close()


def test_transcript_repr_set_between_batches_reaches_every_member():
    """zk_pk_set_transcript_repr AFTER a batch has made the key's member records (by-value copies of the key record): the next
    batch's proofs j > 0 must hash the new value too — every proof equals zk_prove under the same key (round-5 advice: the
    members kept the value they were copied with, and proofs 1 .. B - 1 silently stopped verifying)."""
    eng = zk.Engine(0)
    B = 3
    pk, sets, asgs, opk = _setup(eng, SHAPES["k17like"], [0x5EED0300 + i for i in range(B)])
    rng_seeds = [bytes([60 + i]) * 32 for i in range(B)]
    first = eng.prove_batch(pk, sets, rng_seeds, E.ZK_TRANSCRIPT_EVM)  # (creates the members)
    assert first == [eng.prove(pk, sets[j], rng_seeds[j], E.ZK_TRANSCRIPT_EVM) for j in range(B)]
    other = np.array([0x1234, 0x5678, 0x9ABC, 0x0DEF], dtype=np.uint64)  # any Montgomery image < r
    eng.pk_set_transcript_repr(pk, other)
    for tr in (E.ZK_TRANSCRIPT_EVM, E.ZK_TRANSCRIPT_BLAKE2B):
        lone = [eng.prove(pk, sets[j], rng_seeds[j], tr) for j in range(B)]
        assert eng.prove_batch(pk, sets, rng_seeds, tr) == lone
        assert lone[0] != first[0]  # (the new value is in the transcript)
    eng.close()
