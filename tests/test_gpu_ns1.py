"""NS-1 closed on the GPU box: the bytes the reference's generated verifier accepts are the bytes THIS build emits.

tests/test_oracle_verifier.py::test_reference_yul_verifier_accepts_engine_proof runs the reference's own
`proving-server/P256Verifier.yul` (in place, CPU container only) on tests/golden/engine_proof_k17_evm.json.  That fixture was
made on a GPU by tests/golden/make_engine_fixture.py; here the same inputs (k = 17 bench shape, witness seed 0x5eed0019,
rng seed bytes(range(32)), EVM transcript + GWC — what `/prove_evm` does, halo2-circuits/src/ecc/ecdsa_p256.rs:366-373) go
through the current engine and must give the fixture's proof bytes and verifying key, so that the chain
"reference Yul accepts <- fixture <- device" has no stale link."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_engine_reproduces_the_yul_accepted_fixture(engine):
    import webauthn_halo2_amd as zk
    from webauthn_halo2_amd import engine as E
    from zkoracle import cops

    d = json.load(open(os.path.join(GOLD, "engine_proof_k17_evm.json")))
    p = zk.circuit.K17
    assert (d["k"], d["num_advice"], d["num_lookup_advice"], d["num_fixed"], d["lookup_bits"]) == (
        p.degree, p.num_advice, p.num_lookup_advice, p.num_fixed, p.lookup_bits)
    assert d["witness_seed"] == "0x5eed0019" and bytes.fromhex(d["rng_seed"]) == bytes(range(32))
    eng = engine
    asg = zk.circuit.synthesize(p, 0x5EED0019)
    eng.srs_setup(p.degree)
    pk = eng.keygen(p, np.stack([asg.to_limbs(c) for c in asg.fixed]), asg.copies)
    polys = []
    try:
        for col in asg.advice:
            h = eng.poly(1 << p.degree)
            polys.append(h)
            eng.upload_canonical(h, asg.to_limbs(col))
        proof = eng.prove(pk, polys, bytes(range(32)), E.ZK_TRANSCRIPT_EVM, E.ZK_SCHEME_GWC)
        fc, pc, tr = eng.vk_export(pk)
    finally:
        for h in polys:
            h.free()
        eng.pk_free(pk)
    assert len(proof) == 2720  # halo2-circuits/src/results/ecdsa_bench.csv:4 (EVM/GWC size of the k = 17 row)
    assert proof.hex() == d["proof"], "the engine no longer emits the bytes the reference Yul verifier was shown to accept"
    assert hex(cops.fr_ints(tr.reshape(1, 4))[0]) == d["transcript_repr"]
    assert [[hex(a), hex(b)] for a, b in cops.affine_arr_to_ints(fc)] == d["fixed_commitments"]
    assert [[hex(a), hex(b)] for a, b in cops.affine_arr_to_ints(pc)] == d["permutation_commitments"]
