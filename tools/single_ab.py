"""Single-proof wall clock at k=19 (Blake2b + SHPLONK), min / median of 10, one pipeline — for A/B builds (ZKMI355_LIB)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import webauthn_halo2_amd as zk
from webauthn_halo2_amd import engine as E

p = zk.circuit.K19
eng = zk.Engine(0)
eng.srs_setup(p.degree)
asg = zk.circuit.synthesize(p, 0x5EED0019)
pk = eng.keygen(p, np.stack([asg.to_limbs(c) for c in asg.fixed]), asg.copies)
h = eng.poly(1 << p.degree)
eng.upload_canonical(h, asg.to_limbs(asg.advice[0]))
for tk, name in ((E.ZK_TRANSCRIPT_BLAKE2B, "blake2b"), (E.ZK_TRANSCRIPT_EVM, "evm")):
    for _ in range(3):
        eng.prove(pk, [h], bytes(32), tk)
    ts = []
    for i in range(12):
        t0 = time.perf_counter(); eng.prove(pk, [h], bytes([i]) * 32, tk); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    print("k19 %s single proof: min %.2f ms  median %.2f ms" % (name, ts[0], ts[len(ts) // 2]), flush=True)
