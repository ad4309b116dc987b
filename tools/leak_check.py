import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, ctypes
import webauthn_halo2_amd as zk
from webauthn_halo2_amd import engine as E
hip = ctypes.CDLL("libamdhip64.so")
def used():
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t))
    return (t.value - f.value) / 2**30
p = zk.circuit.K19 if len(sys.argv) > 1 and sys.argv[1] == "19" else zk.circuit.K17
asg = zk.circuit.synthesize(p, 1)
fx = np.stack([asg.to_limbs(c) for c in asg.fixed])
base = used()
for it in range(6):
    eng = zk.Engine(0)
    eng.srs_setup(p.degree)
    pk = eng.keygen(p, fx, asg.copies)
    polys = []
    for col in asg.advice:
        h = eng.poly(1 << p.degree); eng.upload_canonical(h, asg.to_limbs(col)); polys.append(h)
    a = eng.prove(pk, polys, bytes(32), E.ZK_TRANSCRIPT_EVM)
    eng.close()
    print("cycle", it, "used GiB after close %.3f" % (used() - base), len(a))
