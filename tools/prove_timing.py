"""Single-proof wall clock for the BASELINE shapes (k=17 server default, k=19) x (Blake2b+SHPLONK,
Keccak+GWC), one proof at a time on one GPU (the proofs themselves are verified in tests/test_gpu_prover.py)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import webauthn_halo2_amd as zk
from webauthn_halo2_amd import engine as E

out = {}
for name, p in (("k17", zk.circuit.K17), ("k19", zk.circuit.K19)):
    eng = zk.Engine(0)
    k = p.degree
    t0 = time.time(); eng.srs_setup(k); t_srs = time.time() - t0
    asg = zk.circuit.synthesize(p, zk.batch.job_seed(0))
    t0 = time.time(); pk = eng.keygen(p, np.stack([asg.to_limbs(c) for c in asg.fixed]), asg.copies); t_key = time.time() - t0
    polys = []
    for col in asg.advice:
        h = eng.poly(1 << k); eng.upload_canonical(h, asg.to_limbs(col)); polys.append(h)
    for tname, tk in (("blake2b_shplonk", E.ZK_TRANSCRIPT_BLAKE2B), ("evm_gwc", E.ZK_TRANSCRIPT_EVM)):
        for _ in range(2):
            eng.prove(pk, polys, bytes(32), tk)
        ts = []
        for i in range(8):
            t0 = time.perf_counter(); pf = eng.prove(pk, polys, i.to_bytes(32, "little"), tk); ts.append((time.perf_counter() - t0) * 1e3)
        out[f"{name}_{tname}"] = {"ms_min": round(min(ts), 2), "ms_median": round(sorted(ts)[len(ts) // 2], 2), "proof_bytes": len(pf)}
    out[f"{name}_setup"] = {"srs_setup_s": round(t_srs, 3), "keygen_s": round(t_key, 3)}
    eng.close()
print(json.dumps(out))
