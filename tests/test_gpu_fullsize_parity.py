"""Byte parity at the BASELINE sizes: proofs of the k=17 and k=19 shapes made on the device equal, byte for
byte, the committed proofs the oracle's CPU prover made for the same witness seed and the same ChaCha20 RNG
stream (tests/golden/fullsize_proofs.json, generator: tests/golden/make_fullsize_fixtures.py) — both
transcripts, both opening schemes, a worst-case (all-uniform) witness; the device keygen's verifying-key
commitments equal the oracle's.  This exercises what the toy sizes cannot: the two-lane column-batched MSM
passes, 3-pass NTT plans, the 13-bit window tables."""
import hashlib
import json
import os

import numpy as np
import pytest

import webauthn_halo2_amd as zk
from webauthn_halo2_amd import engine as E
from zkoracle import cops, field as F

pytestmark = pytest.mark.gpu

FIX = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize_proofs.json")))
KIND = {"evm": E.ZK_TRANSCRIPT_EVM, "blake2b": E.ZK_TRANSCRIPT_BLAKE2B}
_keys = {}


def key_for(engine, fx):
    """One SRS + proving key per degree, shared by the cases of that degree (the SRS reload between degrees
    invalidates the other key: it is rebuilt when needed)."""
    k = fx["degree"]
    if _keys.get("k") != k:
        for h in _keys.get("polys", []):
            h.free()
        _keys.clear()
        p = zk.circuit.CircuitParams(degree=k, num_advice=fx["num_advice"], num_lookup_advice=fx["num_lookup_advice"],
                                     num_fixed=fx["num_fixed"], lookup_bits=fx["lookup_bits"])
        asg = zk.circuit.synthesize(p, 0)
        engine.srs_setup(k)
        _keys.update(k=k, p=p, pk=engine.keygen(p, np.stack([asg.to_limbs(c) for c in asg.fixed]), asg.copies), polys=[])
    return _keys["p"], _keys["pk"]


@pytest.mark.parametrize("name", sorted(FIX))
def test_fullsize_proof_bytes_equal_the_committed_oracle_proof(engine, name):
    fx = FIX[name]
    p, pk = key_for(engine, fx)
    fc, pc, tr = engine.vk_export(pk)
    as_hex = lambda pts: [["%064x" % c for c in pt] for pt in cops.affine_arr_to_ints(pts)]
    assert as_hex(fc) == fx["vk_fixed_commitments"] and as_hex(pc) == fx["vk_permutation_commitments"]
    assert "%064x" % cops.fr_ints(tr.reshape(1, 4))[0] == fx["transcript_repr"]
    asg = zk.circuit.synthesize(p, fx["witness_seed"], worst_case=fx["worst_case"])
    polys = []
    for col in asg.advice:
        h = engine.poly(1 << fx["degree"])
        engine.upload_canonical(h, asg.to_limbs(col))
        polys.append(h)
    proof = engine.prove(pk, polys, bytes.fromhex(fx["rng_seed"]), KIND[fx["transcript"]])
    for h in polys:
        h.free()
    assert len(proof) == fx["proof_len"]
    assert hashlib.sha256(proof).hexdigest() == fx["sha256"]
    assert proof.hex() == fx["proof"]


@pytest.mark.parametrize("log_n", [19, 21])
def test_ntt_full_output_at_baseline_sizes(engine, log_n):
    """Every one of the 2^19 / 2^21 outputs of the device NTT (forward and inverse, through the fine-grained seam)
    against the oracle's C restatement of best_fft."""
    n = 1 << log_n
    a = np.frombuffer(np.random.default_rng(log_n).bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
    a[:, 3] &= 0x0FFFFFFFFFFFFFFF
    w = F.omega(log_n)
    assert np.array_equal(engine.ntt(a, cops.fr_mont([w])[0], log_n), cops.ntt(a, w, log_n))
    wi = F.inv(w, F.R)
    assert np.array_equal(engine.ntt(a, cops.fr_mont([wi])[0], log_n), cops.ntt(a, wi, log_n))
