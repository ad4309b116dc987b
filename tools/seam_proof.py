"""The literal drop-in's engine time per k = 19 proof (bench.py's single_proof_seam_ms) and its parts: 12 zk_msm_srs, 5 zk_ntt_bn254_fr
at 2^19, 6 at 2^21 — host buffers in and out on every call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import webauthn_halo2_amd as zk
from webauthn_halo2_amd import engine as E
import bench

eng = zk.Engine(0)
eng.srs_setup(19)
print("single_proof_seam_ms %.1f" % bench.seam_single_proof_ms(eng))
n = 1 << 19
rng = np.random.default_rng(3)
s = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64); s[:, 3] &= np.uint64((1 << 60) - 1)
R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
for k in (19, 20, 21):
    a = np.ascontiguousarray(np.tile(s, (1 << (k - 19), 1)))
    w = pow(pow(7, (R - 1) >> 28, R), 1 << (28 - k), R)
    wm = np.frombuffer(((w << 256) % R).to_bytes(32, "little"), dtype=np.uint64).copy()
    ts = []
    for _ in range(6):
        t0 = time.perf_counter(); eng._chk(eng.L.zk_ntt_bn254_fr(eng.ctx, E._p(a), E._p(wm), k), "ntt"); ts.append((time.perf_counter() - t0) * 1e3)
    print("zk_ntt_bn254_fr 2^%d: %.2f ms (min of 6; %d MiB each way)" % (k, min(ts[1:]), 32 << (k - 19) >> 1))
ts = []
for _ in range(6):
    t0 = time.perf_counter(); eng.msm_srs(s, 1); ts.append((time.perf_counter() - t0) * 1e3)
print("zk_msm_srs 2^19: %.2f ms" % min(ts[1:]))
