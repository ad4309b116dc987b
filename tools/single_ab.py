"""Single-proof wall clock at k=19 (Blake2b + SHPLONK), min / median of 10, one pipeline — for A/B builds (ZKMI355_LIB)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import webauthn_halo2_amd as zk
from webauthn_halo2_amd import engine as E

p = zk.circuit.K17 if os.environ.get("K") == "17" else (zk.circuit.CircuitParams(degree=18, num_advice=2, num_lookup_advice=1, num_fixed=1, lookup_bits=17) if os.environ.get("K") == "18" else zk.circuit.K19)
eng = zk.Engine(0)
if os.environ.get("MSM_WINDOW"):
    eng.set_option(E.ZK_OPT_MSM_WINDOW, int(os.environ["MSM_WINDOW"]))
for o in os.environ.get("OPTS", "").split(","):  # OPTS=8=2,5=1: zk_ctx_set_option(id, value)
    if o:
        eng.set_option(*[int(x) for x in o.split("=")])
eng.srs_setup(p.degree)
asg = zk.circuit.synthesize(p, 0x5EED0019)
pk = eng.keygen(p, np.stack([asg.to_limbs(c) for c in asg.fixed]), asg.copies)
hs = []
for col in asg.advice:
    h = eng.poly(1 << p.degree)
    eng.upload_canonical(h, asg.to_limbs(col))
    hs.append(h)
for tk, name in ((E.ZK_TRANSCRIPT_BLAKE2B, "blake2b"), (E.ZK_TRANSCRIPT_EVM, "evm")):
    for _ in range(3):
        eng.prove(pk, hs, bytes(32), tk)
    ts = []
    for i in range(12):
        t0 = time.perf_counter(); eng.prove(pk, hs, bytes([i]) * 32, tk); ts.append((time.perf_counter() - t0) * 1e3)
    ts.sort()
    proof = eng.prove(pk, hs, bytes(32), tk)
    import hashlib
    print("k%d opts %s sha %s" % (p.degree, os.environ.get("OPTS", "-"), hashlib.sha256(proof).hexdigest()[:16]), end="  ")
    print("k%d window %s %s single proof: min %.2f ms  median %.2f ms" % (p.degree, os.environ.get("MSM_WINDOW", "auto"), name, ts[0], ts[len(ts) // 2]), flush=True)
