"""Aggregate rate of lone-column commitments (2^19, resident SRS, uniformly random scalars) over P contexts of one device, each
on its own host thread: how close P overlapping MSM passes come to the accumulation's solo rate (0.57 ms per column).
usage: commit_scale.py [P ...]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import webauthn_halo2_amd as zk

k, reps = 19, 200
n = 1 << k
a = np.frombuffer(np.random.default_rng(1).bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
a[:, 3] &= 0x0FFFFFFFFFFFFFFF
first = zk.Engine(0)
first.srs_setup(k)
for P in [int(x) for x in (sys.argv[1:] or ["1", "2", "3", "4", "6", "8"])]:
    engs = [first] + [zk.Engine(0, share_with=first) for _ in range(P - 1)]
    cols = [e.poly(n, a) for e in engs]
    for e, c in zip(engs, cols):
        e.commit(c, 1)

    def work(e, c):
        for _ in range(reps):
            e.commit(c, 1)

    ths = [threading.Thread(target=work, args=(e, c)) for e, c in zip(engs, cols)]
    t0 = time.perf_counter()
    [t.start() for t in ths]
    [t.join() for t in ths]
    dt = time.perf_counter() - t0
    print("P = %d contexts: %.0f commitments/s = %.3f ms per commitment (aggregate)" % (P, P * reps / dt, dt / (P * reps) * 1e3), flush=True)
    for c in cols:
        c.free()
    for e in engs[1:]:
        e.close()
