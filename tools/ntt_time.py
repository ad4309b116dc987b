"""NTT operator times on the GPU box (HIP events on the engine's stream, best of 8): the transforms a k = 19 proof makes.
Run under a variant build with ZKMI355_LIB=... (tools/ab_variants.sh) for A/B comparisons."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import webauthn_halo2_amd as zk

eng = zk.Engine(0)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 19
n = 1 << k
a = np.frombuffer(np.random.default_rng(1).bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
a[:, 3] &= 0x0FFFFFFFFFFFFFFF
p = eng.poly(n, a)
ext = eng.poly(4 * n)
big = eng.poly(4 * n, np.concatenate([a] * 4))


def best(fn, reps=8):
    ts = []
    for _ in range(reps):
        fn(); eng.sync(); ts.append(eng.last_ms(1))
    return min(ts)


print("lagrange_to_coeff 2^%d      %.4f ms" % (k, best(lambda: eng.lagrange_to_coeff(p))))
print("coeff_to_lagrange 2^%d      %.4f ms" % (k, best(lambda: eng.coeff_to_lagrange(p))))
print("coeff_to_extended 2^%d->2^%d %.4f ms" % (k, k + 2, best(lambda: eng.coeff_to_extended(p, ext))))
print("coeff_to_lagrange 2^%d      %.4f ms" % (k + 2, best(lambda: eng.coeff_to_lagrange(big))))
print("lagrange_to_coeff 2^%d      %.4f ms" % (k + 2, best(lambda: eng.lagrange_to_coeff(big))))
print("extended_to_coeff 2^%d      %.4f ms" % (k + 2, best(lambda: eng.extended_to_coeff(big, 3 * n))))
