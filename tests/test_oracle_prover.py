"""The oracle's restated create_proof makes proofs that the golden-proof-pinned
verifier accepts, for every BASELINE shape (scaled to small k), both transcripts.
CPU only."""
import pytest

import webauthn_halo2_amd as zk
from zkoracle import plonk, prover
from zkoracle.hashes import ChaCha20Rng

SHAPES = {  # (A, L, F, k, lookup_bits): k=19-, k=17-, k=18-like, and one with F=2 / L=2
    "k19like": (1, 1, 1, 7, 6),
    "k17like": (4, 1, 1, 7, 5),
    "k18like": (2, 1, 1, 6, 4),
    "wide": (3, 2, 2, 7, 5),
    "idle": (4, 1, 1, 6, 4, 1),   # last gate column never enabled: no selector column (bench rows k <= 13)
}


def build(name, seed=0x5EED0019, worst=False):
    A, L, F, k, lb, idle = (SHAPES[name] + (0,))[:6]
    p = zk.circuit.CircuitParams(degree=k, num_advice=A, num_lookup_advice=L, num_fixed=F, lookup_bits=lb,
                                 idle_gate_columns=idle)
    asg = zk.circuit.synthesize(p, seed, worst_case=worst)
    sh = plonk.Shape(k, A, L, F, lb, idle)
    assert sh.perm_cols == asg.layout.perm_cols and sh.n_fix == asg.layout.n_fix
    return prover.keygen(prover.Circuit(sh, asg.fixed, asg.copies, asg.advice)), asg


@pytest.mark.parametrize("name", list(SHAPES))
@pytest.mark.parametrize("kind", ["evm", "blake2b"])
def test_proof_roundtrip(name, kind):
    pk, asg = build(name)
    proof = prover.create_proof(pk, asg.advice, ChaCha20Rng(b"\x07" * 32), kind)
    sh = pk.shape
    pts = sh.n_points_before_multiopen()
    if kind == "evm":
        assert len(proof) == 64 * (pts + sh.gwc_sets()) + 32 * sh.n_evals()
    else:
        assert len(proof) == 32 * (pts + 2 + sh.n_evals())
    assert plonk.verify(pk.vk, proof, kind)
    for pos in (0, len(proof) // 3, len(proof) - 1):
        bad = bytearray(proof)
        bad[pos] ^= 4
        assert not plonk.verify(pk.vk, bytes(bad), kind)
    # same RNG stream -> same bytes; different stream -> different proof, still valid
    assert proof == prover.create_proof(pk, asg.advice, ChaCha20Rng(b"\x07" * 32), kind)
    other = prover.create_proof(pk, asg.advice, ChaCha20Rng(b"\x08" * 32), kind)
    assert other != proof and plonk.verify(pk.vk, other, kind)


def test_unsatisfied_witness_is_caught():
    pk, asg = build("k19like")
    adv = [list(c) for c in asg.advice]
    adv[0][3] = (adv[0][3] + 1) % zk.circuit.R  # break gate 0: d != a + b*c
    # degree-5 shape: h fills the whole extended domain, so the prover cannot notice; the verifier does
    proof = prover.create_proof(pk, adv, ChaCha20Rng(bytes(32)), "evm")
    assert not plonk.verify(pk.vk, proof, "evm")
    # degree-4 shapes: the quotient spills past (d-1)*n coefficients and the prover itself asserts
    pk, asg = build("k17like")
    adv = [list(c) for c in asg.advice]
    adv[1][3] = (adv[1][3] + 1) % zk.circuit.R
    with pytest.raises(AssertionError):
        prover.create_proof(pk, adv, ChaCha20Rng(bytes(32)), "evm")


def test_cross_transcript_schemes():
    # GWC over Blake2b and SHPLONK over Keccak also verify (the four Prover/Transcript pairings)
    pk, asg = build("k18like")
    for kind, scheme in (("blake2b", "gwc"), ("evm", "shplonk")):
        proof = prover.create_proof(pk, asg.advice, ChaCha20Rng(bytes(32)), kind, scheme)
        assert plonk.verify(pk.vk, proof, kind, scheme)
        assert not plonk.verify(pk.vk, proof, kind, "gwc" if scheme == "shplonk" else "shplonk")


def test_published_proof_sizes_of_every_bench_row():
    """K6 widened: the shape model gives the published Blake2b/SHPLONK proof size of all nine rows of
    halo2-circuits/src/results/ecdsa_bench.csv; rows k <= 13 need 1/2/3 idle gate columns."""
    rows = [(19, 1, 1, 1, 18, 0, 960), (18, 2, 1, 1, 17, 0, 1344), (17, 4, 1, 1, 16, 0, 1920),
            (16, 8, 2, 1, 15, 0, 3552), (15, 17, 3, 1, 14, 0, 6560), (14, 34, 6, 1, 13, 0, 12704),
            (13, 68, 12, 1, 12, 1, 24960), (12, 139, 24, 2, 11, 2, 50496), (11, 291, 53, 4, 10, 3, 106496)]
    for k, A, L, F, lb, idle, size in rows:
        sh = plonk.Shape(k, A, L, F, lb, idle)
        assert 32 * (sh.n_points_before_multiopen() + 2 + sh.n_evals()) == size, k
