"""Runs each hot operator a few times on resident data (for rocprofv3 --kernel-trace --stats)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import webauthn_halo2_amd as zk

k = int(sys.argv[1]) if len(sys.argv) > 1 else 19
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n = 1 << k
eng = zk.Engine(0)
t = time.time(); eng.srs_setup(k); print("srs_setup k=%d: %.3f s" % (k, time.time() - t))
a = np.frombuffer(np.random.default_rng(1).bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
a[:, 3] &= 0x0FFFFFFFFFFFFFFF
p = eng.poly(n, a)
ext = eng.poly(4 * n)
for basis in (0, 1):
    for _ in range(reps):
        t = time.time(); eng.commit(p, basis); wall = time.time() - t
    print("commit basis=%d: gpu %.3f ms, wall %.3f ms" % (basis, eng.last_ms(0), wall * 1e3))
for _ in range(reps):
    t = time.time(); eng.lagrange_to_coeff(p); eng.sync(); wall = time.time() - t
print("intt 2^%d: gpu %.3f ms wall %.3f ms" % (k, eng.last_ms(1), wall * 1e3))
for _ in range(reps):
    t = time.time(); eng.coeff_to_extended(p, ext); eng.sync(); wall = time.time() - t
print("coset ntt 2^%d: gpu %.3f ms wall %.3f ms" % (k + 2, eng.last_ms(1), wall * 1e3))
for _ in range(reps):
    t = time.time(); eng.extended_to_coeff(ext, 4 * n); eng.sync(); wall = time.time() - t
print("coset intt 2^%d: gpu %.3f ms wall %.3f ms" % (k + 2, eng.last_ms(1), wall * 1e3))
x = a[5]
for _ in range(reps):
    t = time.time(); eng.eval(p, x); wall = time.time() - t
print("eval 2^%d: gpu %.3f ms wall %.3f ms" % (k, eng.last_ms(3), wall * 1e3))
# witness-like scalars
rng = np.random.default_rng(2)
small = np.zeros((n, 4), dtype=np.uint64)
small[:, 0] = rng.integers(0, 1 << 18, n)
ps = eng.poly(n, small)  # (not Montgomery-converted: just a skewed distribution of residues)
for _ in range(2):
    eng.commit(ps, 1)
print("commit skewed: gpu %.3f ms" % eng.last_ms(0))
