"""Head / accumulate / tail of a lone 2^19 commitment and of a 2-column pass, HIP events (ZK_T_MSM, ZK_T_MSM_ACCUM, ZK_T_MSM_TAIL), for
A/B builds (ZKMI355_LIB): tools/msm_parts.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import webauthn_halo2_amd as zk
from webauthn_halo2_amd import engine as E

k = int(os.environ.get("K", "19"))
n = 1 << k
rng = np.random.default_rng(7)
eng = zk.Engine(0)
for o in os.environ.get("OPTS", "").split(","):  # OPTS=10=1,5=2: zk_ctx_set_option before the SRS is loaded
    if o:
        eng.set_option(*[int(x) for x in o.split("=")])
eng.srs_setup(k)
cols = []
for i in range(2):
    a = np.frombuffer(rng.bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
    a[:, 3] &= 0x0FFFFFFFFFFFFFFF
    cols.append(eng.poly(n, a))
want = [eng.commit(c, 1).copy() for c in cols]
tag = os.path.basename(os.environ.get("ZKMI355_LIB", "base")) + (" OPTS=" + os.environ["OPTS"] if os.environ.get("OPTS") else "")
for cnt in (1, 2):
    eng.timer_reset()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        got = eng.commit_batch(cols[:cnt], 1)
    wall = (time.perf_counter() - t0) / reps * 1e3
    assert all(np.array_equal(got[j], want[j]) for j in range(cnt))
    msm, m = eng.timer_stats(E.ZK_T_MSM)
    acc, a_ = eng.timer_stats(E.ZK_T_MSM_ACCUM)
    tail, t_ = eng.timer_stats(E.ZK_T_MSM_TAIL)
    print("%-28s %d column(s): head %.3f  accumulate %.3f  tail %.3f  wall %.3f ms per pass" %
          (tag, cnt, (msm - acc) / max(m, 1), acc / max(a_, 1), tail / max(t_, 1), wall), flush=True)
