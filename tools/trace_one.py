"""Proves one bench_ecdsa.config row repeatedly (for rocprofv3 --kernel-trace + tools/timeline.py).
usage: trace_one.py <K19|K17|k,A,L,F,lookup_bits[,idle]> <blake2b|evm> [reps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import webauthn_halo2_amd as zk
from webauthn_halo2_amd import engine as E

spec, kind = sys.argv[1], sys.argv[2]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
if spec in ("K19", "K17"):
    p = getattr(zk.circuit, spec)
else:
    v = [int(x) for x in spec.split(",")] + [0]
    p = zk.circuit.CircuitParams(degree=v[0], num_advice=v[1], num_lookup_advice=v[2], num_fixed=v[3], lookup_bits=v[4],
                                 idle_gate_columns=v[5])
eng = zk.Engine(0)
if os.environ.get("MSM_WINDOW", "0") != "0":
    eng.set_option(E.ZK_OPT_MSM_WINDOW, int(os.environ["MSM_WINDOW"]))
eng.srs_setup(p.degree)
asg = zk.circuit.synthesize(p, 0x5EED0019)
pk = eng.keygen(p, np.stack([asg.to_limbs(c) for c in asg.fixed]), asg.copies)
polys = []
for col in asg.advice:
    h = eng.poly(1 << p.degree)
    eng.upload_canonical(h, asg.to_limbs(col))
    polys.append(h)
tk = E.ZK_TRANSCRIPT_EVM if kind == "evm" else E.ZK_TRANSCRIPT_BLAKE2B
ts = []
for i in range(reps):
    t0 = time.perf_counter()
    pf = eng.prove(pk, polys, bytes([i + 1]) * 32, tk)
    ts.append((time.perf_counter() - t0) * 1e3)
print(spec, kind, "bytes", len(pf), "ms min %.2f median %.2f" % (min(ts), sorted(ts)[len(ts) // 2]))
