"""BASELINE.json configs[4]: k=21 large-SRS stress — 2^21-point BN254 G1 MSM + 2^21 NTT on one
MI355X, timed with HIP events and self-checked without the test oracle (commit(p) == commit_lagrange(NTT p),
NTT round trip; the tau-oracle comparison is tests/test_gpu_ops.py::test_k21_stress_msm_tau_oracle); run under
rocprofv3 for the HBM counters.  Prints one JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import webauthn_halo2_amd as zk

K = 21
n = 1 << K
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
check = "--no-check" not in sys.argv
eng = zk.Engine(0)
t0 = time.time(); eng.srs_setup(K); t_srs = time.time() - t0
s = np.frombuffer(np.random.default_rng(0x5EED0021).bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
s[:, 3] &= 0x0FFFFFFFFFFFFFFF
p = eng.poly(n, s)
ms_msm, ms_acc = [], []
for _ in range(reps):
    c = eng.commit(p, 0)
    ms_msm.append(eng.last_ms(0)); ms_acc.append(eng.last_ms(4))
if check:
    v = eng.poly(n, s)
    eng.coeff_to_lagrange(v)
    assert (eng.commit(v, 1) == c).all(), "commit(p) != commit_lagrange(NTT p) at 2^21"
    v.free()
q = eng.poly(n, s)
ms_ntt = []
for _ in range(reps):
    eng.coeff_to_lagrange(q); eng.sync(); ms_ntt.append(eng.last_ms(1))
    eng.lagrange_to_coeff(q); eng.sync(); ms_ntt.append(eng.last_ms(1))
if check:
    assert np.array_equal(eng.download(q), s), "iNTT(NTT(x)) != x at 2^21"
msm, acc, ntt = min(ms_msm), min(ms_acc), min(ms_ntt)
print(json.dumps({
    "config": "k=21 stress (BASELINE configs[4])", "n": n, "srs_setup_s": round(t_srs, 3),
    "msm_head_ms": msm, "msm_accumulate_ms": acc, "ntt_ms": ntt,
    "msm_algorithmic_GBps": 96.0 * n / (acc * 1e-3) / 1e9, "ntt_algorithmic_GBps": 64.0 * n / (ntt * 1e-3) / 1e9,
    "hbm_peak_GBps": 8000, "checked": check}))
