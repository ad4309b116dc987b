#!/usr/bin/env python3
"""Generates tests/golden/batch_k19_sha256.json: the SHA-256 of the proof of EVERY job of BASELINE.json configs[3] (256 independent
k = 19 proofs; bench.py's `--gpus 8 --steps 32` is exactly this batch) as made by the oracle's CPU prover
(oracle/zkoracle/fastprover.py): job j = witness seed 0x5eed0019 + j (batch.job_seed), blinding stream batch.job_rng_seed(j),
Blake2b + SHPLONK, 960 bytes.  bench.py compares the digest of every timed proof with this table (data only: no oracle code
runs on the GPU box), tests/test_gpu_prover.py proves a slice of the batch through four pipelines and compares.

Run in the build container (about 15 s per job on 8 cores; resumable: jobs already in the file are skipped):
    python tests/golden/make_batch_hashes.py [first [count]]
    python tests/golden/make_batch_hashes.py 0 40 k17evm      -> batch_k17_evm_sha256.json
Round 6: `k17evm` makes the same table for the PROVING SERVER's configuration (k = 17, EVM transcript + GWC, 2 720-byte proofs,
proving-server/src/main.rs:17,64-79): the 40 jobs bench.py's `k17_evm_proofs_per_sec` proves and compares.
Only expected outputs are stored; the inputs are regenerated from the seeds."""
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

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "batch_k19_sha256.json")


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    k17 = len(sys.argv) > 3 and sys.argv[3] == "k17evm"
    global OUT
    p = zk.circuit.K17 if k17 else zk.circuit.K19
    kind, size = ("evm", 2720) if k17 else ("blake2b", 960)
    if k17:
        OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "batch_k17_evm_sha256.json")
    sh = plonk.Shape(p.degree, p.num_advice, p.num_lookup_advice, p.num_fixed, p.lookup_bits)
    out = json.load(open(OUT)) if os.path.exists(OUT) else {"degree": p.degree, "transcript": kind, "multiopen": "gwc" if k17 else "shplonk", "sha256": {}}
    asg0 = zk.circuit.synthesize(p, 0)
    pk = fp.keygen(sh, asg0.fixed, asg0.copies)
    for j in range(first, first + count):
        if str(j) in out["sha256"]:
            continue
        t0 = time.time()
        asg = zk.circuit.synthesize(p, batch.job_seed(j))
        proof = fp.create_proof(pk, asg.advice, ChaCha20Rng(batch.job_rng_seed(j)), kind)
        assert len(proof) == size
        if j % 16 == 0:
            assert plonk.verify(pk.vk, proof, kind), j
        out["sha256"][str(j)] = hashlib.sha256(proof).hexdigest()
        with open(OUT + ".tmp", "w") as f:
            json.dump(out, f, indent=0, sort_keys=True)
        os.replace(OUT + ".tmp", OUT)
        print("job %d  sha256 %s  (%.1f s)" % (j, out["sha256"][str(j)][:16], time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
