import sys,os,time
sys.path.insert(0, os.getcwd())
import numpy as np, webauthn_halo2_amd as zk
from webauthn_halo2_amd import engine as E
# usage: k17_timing.py [msm_window_bits]
p=zk.circuit.K17; eng=zk.Engine(0)
win=int(sys.argv[1]) if len(sys.argv)>1 else 0
eng.set_option(E.ZK_OPT_MSM_WINDOW, win); eng.srs_setup(17)
asg=zk.circuit.synthesize(p,1); pk=eng.keygen(p,np.stack([asg.to_limbs(c) for c in asg.fixed]),asg.copies)
polys=[]
for col in asg.advice:
    h=eng.poly(1<<17); eng.upload_canonical(h,asg.to_limbs(col)); polys.append(h)
for tk in (E.ZK_TRANSCRIPT_BLAKE2B, E.ZK_TRANSCRIPT_EVM):
    for i in range(3): eng.prove(pk,polys,bytes(32),tk)
    ts=[]
    for i in range(8):
        t0=time.perf_counter(); eng.prove(pk,polys,bytes(32),tk); ts.append((time.perf_counter()-t0)*1e3)
    print("window", win or "auto", "transcript", tk, round(min(ts),2))
