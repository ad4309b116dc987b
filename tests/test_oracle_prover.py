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
}


def build(name, seed=0x5EED0019, worst=False):
    A, L, F, k, lb = SHAPES[name]
    p = zk.circuit.CircuitParams(degree=k, num_advice=A, num_lookup_advice=L, num_fixed=F, lookup_bits=lb)
    asg = zk.circuit.synthesize(p, seed, worst_case=worst)
    sh = plonk.Shape(k, A, L, F, lb)
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
