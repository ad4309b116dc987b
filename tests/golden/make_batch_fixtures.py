#!/usr/bin/env python3
"""Generates tests/golden/batch_k19_proofs.json: the first JOBS jobs of the k = 19 batch workload (BASELINE.json configs[3];
bench.py's timed steps) proven by the oracle's CPU prover (oracle/zkoracle/fastprover.py): job j = witness seed
0x5eed0019 + j (batch.job_seed), blinding stream batch.job_rng_seed(j), Blake2b + SHPLONK, 960 bytes each.
tests/test_gpu_prover.py drains the same jobs through batch.run on the device and compares the bytes.

Run in the build container (about 2 minutes on 8 cores):  python tests/golden/make_batch_fixtures.py
Only expected outputs are stored (proof hex + sha256); the inputs are regenerated from the seeds by the test."""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)

import webauthn_halo2_amd as zk  # noqa: E402  (witness generator and job seeds only; no engine is touched)
from webauthn_halo2_amd import batch  # noqa: E402
from zkoracle import fastprover as fp, plonk  # noqa: E402
from zkoracle.hashes import ChaCha20Rng  # noqa: E402

JOBS = 8


def main():
    p = zk.circuit.K19
    sh = plonk.Shape(p.degree, p.num_advice, p.num_lookup_advice, p.num_fixed, p.lookup_bits)
    asg0 = zk.circuit.synthesize(p, 0)
    pk = fp.keygen(sh, asg0.fixed, asg0.copies)
    out = {"degree": p.degree, "transcript": "blake2b", "jobs": {}}
    for j in range(JOBS):
        t0 = time.time()
        asg = zk.circuit.synthesize(p, batch.job_seed(j))
        proof = fp.create_proof(pk, asg.advice, ChaCha20Rng(batch.job_rng_seed(j)), "blake2b")
        assert plonk.verify(pk.vk, proof, "blake2b"), j
        out["jobs"][str(j)] = {"sha256": hashlib.sha256(proof).hexdigest(), "proof": proof.hex()}
        print("job %d  %d bytes  sha256 %s  (%.1f s)" % (j, len(proof), out["jobs"][str(j)]["sha256"][:16], time.time() - t0), flush=True)
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "batch_k19_proofs.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
