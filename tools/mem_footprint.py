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
for name in ("K17", "K19"):
    p = getattr(zk.circuit, name)
    u0 = used()
    eng = zk.Engine(0)
    eng.srs_setup(p.degree); u1 = used()
    asg = zk.circuit.synthesize(p, 1)
    pk = eng.keygen(p, np.stack([asg.to_limbs(c) for c in asg.fixed]), asg.copies); u2 = used()
    polys = []
    for col in asg.advice:
        h = eng.poly(1 << p.degree); eng.upload_canonical(h, asg.to_limbs(col)); polys.append(h)
    eng.prove(pk, polys, bytes(32), E.ZK_TRANSCRIPT_EVM); eng.prove(pk, polys, bytes(32), E.ZK_TRANSCRIPT_BLAKE2B); u3 = used()
    print(name, "GiB: srs+tables %.2f  key+workspace %.2f  msm lanes etc %.2f  total %.2f" % (u1 - u0, u2 - u1, u3 - u2, u3 - u0))
    eng.close()
