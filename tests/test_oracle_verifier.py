"""K5/K3 pin: the oracle's generic PLONK verifier, fed the k=17
verifying key constants baked into the reference's generated verifier, accepts
the reference's golden EVM proof and reproduces its Fiat-Shamir challenges.
CPU only."""
import json
import os

import pytest

from zkoracle import plonk

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")


def load_k17():
    d = json.load(open(os.path.join(GOLD, "vk_k17.json")))
    shape = plonk.Shape(k=17, num_advice=4, num_lookup_advice=1, num_fixed=1, lookup_bits=16)
    pt = lambda p: (int(p[0], 16), int(p[1], 16))
    vk = plonk.VerifyingKey(shape, [pt(p) for p in d["fixed_commitments"]],
                            [pt(p) for p in d["permutation_commitments"]], int(d["transcript_repr"], 16))
    proof = bytes.fromhex(open(os.path.join(GOLD, "golden_proof_k17_evm.hex")).read().strip())
    return d, vk, proof


def test_shape_model_matches_published_proof_sizes():
    # K6: halo2-circuits/src/results/ecdsa_bench.csv:2-7 (Blake2b sizes) and the 2720-byte EVM proof
    rows = {19: (1, 1, 1, 960), 18: (2, 1, 1, 1344), 17: (4, 1, 1, 1920), 16: (8, 2, 1, 3552),
            15: (17, 3, 1, 6560), 14: (34, 6, 1, 12704)}
    for k, (A, L, F, size) in rows.items():
        sh = plonk.Shape(k, A, L, F)
        pts = sh.n_points_before_multiopen() + 2  # SHPLONK: 2 more points
        assert 32 * (pts + sh.n_evals()) == size, k
    sh = plonk.Shape(17, 4, 1, 1)
    assert 64 * (sh.n_points_before_multiopen() + sh.gwc_sets()) + 32 * sh.n_evals() == 2720
    sh = plonk.Shape(19, 1, 1, 1)
    assert 64 * (sh.n_points_before_multiopen() + sh.gwc_sets()) + 32 * sh.n_evals() == 1536


def test_golden_proof_accepted_and_challenges_match():
    d, vk, proof = load_k17()
    ok, pf = plonk.verify(vk, proof, "evm", return_detail=True)
    assert ok
    for name, val in d["golden_challenges"].items():
        assert pf.challenges[name] == int(val, 16), name


def test_tampered_golden_proof_rejected():
    _, vk, proof = load_k17()
    for pos in (5, 0x1c0 + 40, 0x3c0 + 7, 0x3c0 + 32 * 20 + 31, 0x920 + 3, len(proof) - 1):
        bad = bytearray(proof)
        bad[pos] ^= 1
        assert not plonk.verify(vk, bytes(bad), "evm")
    assert not plonk.verify(vk, proof[:-32], "evm")
    assert not plonk.verify(vk, proof + b"\x00" * 32, "evm")
    assert not plonk.verify(vk, b"", "evm")  # contracts/test/P256Account.t.sol:106-118: empty proof must fail


@pytest.mark.skipif(not os.path.exists("/root/reference/proving-server/P256Verifier.yul"),
                    reason="reference tree not present (GPU box)")
def test_reference_yul_verifier_agrees():
    """Runs the reference's generated verifier where it lies through oracle/tools/yul_exec.py."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle", "tools"))
    import yul_exec
    _, vk, proof = load_k17()
    ok, vm = yul_exec.run_verifier("/root/reference/proving-server/P256Verifier.yul", proof)
    assert ok and vm.precompile_calls == {5: 1, 6: 38, 7: 39, 8: 1}
    bad = bytearray(proof)
    bad[777] ^= 0x10
    assert yul_exec.run_verifier("/root/reference/proving-server/P256Verifier.yul", bytes(bad))[0] == plonk.verify(vk, bytes(bad), "evm") == False  # noqa: E712


def load_engine_fixture():
    d = json.load(open(os.path.join(GOLD, "engine_proof_k17_evm.json")))
    shape = plonk.Shape(d["k"], d["num_advice"], d["num_lookup_advice"], d["num_fixed"], d["lookup_bits"])
    pt = lambda p: (int(p[0], 16), int(p[1], 16))
    vk = plonk.VerifyingKey(shape, [pt(p) for p in d["fixed_commitments"]],
                            [pt(p) for p in d["permutation_commitments"]], int(d["transcript_repr"], 16))
    return vk, bytes.fromhex(d["proof"])


def test_engine_proof_fixture_verifies():
    """A proof the device prover made on an MI355X for the k=17 bench shape
    (tests/golden/make_engine_fixture.py) is accepted by the golden-proof-pinned verifier."""
    vk, proof = load_engine_fixture()
    assert len(proof) == 2720
    assert plonk.verify(vk, proof, "evm")
    # the digest the device stamped on its key is halo2's own: the hash of the pinned verifying key's Debug rendering
    # (zkoracle/vkrepr.py, pinned by the reference's k = 17 literal in tests/test_oracle_kat.py)
    from zkoracle import vkrepr
    assert vk.transcript_repr == vkrepr.transcript_repr(vk.shape, vk.fixed_commitments, vk.permutation_commitments)


@pytest.mark.skipif(not os.path.exists("/root/reference/proving-server/P256Verifier.yul"),
                    reason="reference tree not present (GPU box)")
def test_reference_yul_verifier_accepts_engine_proof():
    """BASELINE config 3: the reference's GENERATED verifier program (proving-server/P256Verifier.yul,
    executed where it lies) accepts the device prover's EVM proof bytes unchanged.  The program is
    vk-specific, so the synthetic circuit's 12 vk commitments and transcript_repr replace the ECDSA
    circuit's literals in memory; every instruction of the verification logic is the reference's."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle", "tools"))
    import yul_exec
    vk, proof = load_engine_fixture()
    key = (vk.transcript_repr, list(vk.fixed_commitments) + list(vk.permutation_commitments))
    yul = "/root/reference/proving-server/P256Verifier.yul"
    ok, vm = yul_exec.run_verifier(yul, proof, vk=key)
    assert ok and vm.precompile_calls == {5: 1, 6: 38, 7: 39, 8: 1}
    # the unmodified program (ECDSA circuit's key) must reject it, and a flipped bit must be rejected with our key
    assert not yul_exec.run_verifier(yul, proof)[0]
    bad = bytearray(proof)
    bad[1500] ^= 1
    assert not yul_exec.run_verifier(yul, bytes(bad), vk=key)[0]
