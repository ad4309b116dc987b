"""Generates tests/golden/engine_proof_k17_evm.json ON THE GPU BOX: an EVM/GWC proof of the k=17
bench shape (the shape proving-server/P256Verifier.yul was generated for) made by the device prover,
with its verifying key.  tests/test_oracle_verifier.py feeds it to the reference's generated verifier
program (vk literals swapped in memory) and to the generic oracle verifier.

  gpurun -- 'python tests/golden/make_engine_fixture.py gpurun_out/engine_proof_k17_evm.json'
then copy the file next to this script.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import webauthn_halo2_amd as zk  # noqa: E402
from webauthn_halo2_amd import engine as E  # noqa: E402
from zkoracle import cops  # noqa: E402  (format conversion only)

p = zk.circuit.K17
eng = zk.Engine(0)
asg = zk.circuit.synthesize(p, 0x5EED0019)
eng.srs_setup(p.degree)
pk = eng.keygen(p, np.stack([asg.to_limbs(c) for c in asg.fixed]), asg.copies)
polys = []
for col in asg.advice:
    h = eng.poly(1 << p.degree)
    eng.upload_canonical(h, asg.to_limbs(col))
    polys.append(h)
seed = bytes(range(32))
proof = eng.prove(pk, polys, seed, E.ZK_TRANSCRIPT_EVM, E.ZK_SCHEME_GWC)
fc, pc, tr = eng.vk_export(pk)
out = {
    "k": p.degree, "num_advice": p.num_advice, "num_lookup_advice": p.num_lookup_advice, "num_fixed": p.num_fixed,
    "lookup_bits": p.lookup_bits, "witness_seed": "0x5eed0019", "rng_seed": seed.hex(),
    "transcript_repr": hex(cops.fr_ints(tr.reshape(1, 4))[0]),
    "fixed_commitments": [[hex(a), hex(b)] for a, b in cops.affine_arr_to_ints(fc)],
    "permutation_commitments": [[hex(a), hex(b)] for a, b in cops.affine_arr_to_ints(pc)],
    "proof": proof.hex(),
}
assert len(proof) == 2720
json.dump(out, open(sys.argv[1], "w"), indent=1)
print("wrote", sys.argv[1], len(proof))
