"""Bucket-accumulation time per 2^19 column against the MSM window (ZK_OPT_MSM_WINDOW): 13 bits (20 additions per
scalar, 4096 buckets) against the wide path's 15 / 16 bits (17 / 16 additions, 16384 / 32768 buckets).  HIP-event
time of the accumulate kernel alone and of the whole head, wall clock of the whole commit (head + reduction tail +
host finish), for one column and for column batches; uniform scalars and the advice-column mix."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import webauthn_halo2_amd as zk
from webauthn_halo2_amd import engine as E

k = int(os.environ.get("K", "19"))
n = 1 << k
rng = np.random.default_rng(7)


def col():
    a = np.frombuffer(rng.bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
    a[:, 3] &= 0x0FFFFFFFFFFFFFFF
    return a


def mix_col():
    a = col()
    u = rng.random(n)
    small = u < 0.4
    a[small, 1:] = 0
    a[small, 0] &= 0x3FFFF
    mid = (u >= 0.4) & (u < 0.75)
    a[mid, 2:] = 0
    a[mid, 1] &= 0xFFFFFF
    a[u >= 0.9] = 0
    return a


ref = None
for bits in [int(x) for x in (sys.argv[1:] or ["13", "15", "16"])]:
    eng = zk.Engine(0)
    eng.set_option(E.ZK_OPT_MSM_WINDOW, bits)
    eng.set_option(E.ZK_OPT_MSM_BATCH, 8)
    eng.srs_setup(k)
    rng = np.random.default_rng(7)
    cols = [eng.poly(n, col()) for _ in range(8)]
    mixp = eng.poly(n, mix_col())
    out = [eng.commit(cols[0], 0).tobytes(), eng.commit(mixp, 1).tobytes()]
    if ref is None:
        ref = out
    assert out == ref, "window %d disagrees with the first window" % bits
    only = os.environ.get("COLS")
    for label, polys in (("1 uniform", cols[:1]), ("2 uniform", cols[:2]), ("4 uniform", cols[:4]), ("8 uniform", cols[:8]), ("1 advice-mix", [mixp])):
        if only and label.split()[0] not in only.split(",") or (only and "mix" in label):
            continue
        for _ in range(2):
            eng.commit_batch(polys, 0)
        eng.timer_reset()
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.commit_batch(polys, 0)
        wall = (time.perf_counter() - t0) / reps * 1e3
        acc_ms, acc_n = eng.timer_stats(E.ZK_T_MSM_ACCUM)
        head_ms, head_n = eng.timer_stats(E.ZK_T_MSM)
        print("window %2d  %-13s accumulate %.3f ms/column  head %.3f ms/column  commit wall %.3f ms/column" %
              (bits, label, acc_ms / acc_n / len(polys), head_ms / head_n / len(polys), wall / len(polys)), flush=True)
    eng.close()
