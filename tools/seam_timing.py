import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import webauthn_halo2_amd as zk
eng = zk.Engine(0)
k = 19; n = 1 << k
eng.srs_setup(k)
g = eng.srs_export(0, 0, n)
bases = eng.srs_export(1, 0, n)   # g_lagrange on the host, as a Rust host holds it
eng.srs_load(k, g, bases)         # the host hands its ParamsKZG arrays over once
rng = np.random.default_rng(1)
s = np.frombuffer(rng.bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy(); s[:, 3] &= 0x0FFFFFFFFFFFFFFF
for i in range(2): eng.msm(s, bases)
ts = []
for i in range(5):
    t0 = time.perf_counter(); r = eng.msm(s, bases); ts.append((time.perf_counter() - t0) * 1e3)
print("zk_msm_bn254 2^19 through the seam (arbitrary bases, both operands uploaded): %.2f ms (min of 5)" % min(ts))
ts = []
for i in range(5):
    t0 = time.perf_counter(); r2 = eng.msm_srs(s, 1); ts.append((time.perf_counter() - t0) * 1e3)
print("zk_msm_srs 2^19 (ParamsKZG::commit_lagrange: scalars uploaded, resident window tables): %.2f ms" % min(ts))
p = eng.poly(n, s)
ts = []
for i in range(5):
    t0 = time.perf_counter(); eng.commit(p, 1); ts.append((time.perf_counter() - t0) * 1e3)
print("zk_commit resident: %.2f ms" % min(ts))
R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
omega = pow(pow(7, (R - 1) >> 28, R), 1 << (28 - k), R)          # halo2curves Fr::ROOT_OF_UNITY ^ (2^(28-k))
w = np.frombuffer(((omega << 256) % R).to_bytes(32, "little"), dtype=np.uint64).copy()  # Montgomery image
ts = []
for i in range(5):
    t0 = time.perf_counter(); eng.ntt(s, w, k); ts.append((time.perf_counter() - t0) * 1e3)
print("zk_ntt_bn254_fr 2^19 through the seam (incl. the binding's input copy): %.2f ms" % min(ts))
