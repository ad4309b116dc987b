"""Diagnostics for zk_prove_batch at full size (k = 19): wide MSM passes of 3 .. 8 columns against lone commitments, then lock-step
batches of 2 .. 6 proofs under several pass widths against zk_prove.  Prints what differs; exit code 1 on any difference."""
import sys
import os
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import webauthn_halo2_amd as zk
from webauthn_halo2_amd import batch, engine as E

bad = 0
k = int(sys.argv[1]) if len(sys.argv) > 1 else 19
p = zk.circuit.K19 if k == 19 else zk.circuit.K17
n = 1 << k
jobs = list(range(8))
wit = batch.synthesize_jobs(p, jobs)
fixed, copies = batch.structure(p)

# ---- (a) MSM passes of many columns
eng = zk.Engine(0)
eng.set_option(E.ZK_OPT_MSM_BATCH, 8)
eng.srs_setup(k)
rng = np.random.default_rng(5)
cols = []
for i in range(8):
    a = np.frombuffer(rng.bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
    a[:, 3] &= 0x0FFFFFFFFFFFFFFF
    if i % 3 == 2:
        a[:, 1:] = 0
        a[:, 0] &= 0x3FFFF
    cols.append(eng.poly(n, a))
lone = [eng.commit(c, 1).copy() for c in cols]
for cnt in (2, 3, 4, 5, 6, 7, 8):
    for rep in range(2):
        got = eng.commit_batch(cols[:cnt], 1)
        diff = [j for j in range(cnt) if not np.array_equal(got[j], lone[j])]
        if diff:
            bad += 1
            print("MSM pass of %d columns (rep %d): columns %s differ" % (cnt, rep, diff))
print("msm passes checked")
for c in cols:
    c.free()
eng.close()

# ---- (b) lock-step batches
for cap in (0, 4, 2, 12):
    pl = batch.Pipeline(0, p, fixed, copies, deterministic_seeds=True)
    if cap:
        pl.eng.set_option(E.ZK_OPT_BATCH_PASS_COLUMNS, cap)
    for j in jobs:
        pl.load(j, wit[j])
    want = {j: pl.prove(j, keep=True) for j in jobs}
    for B in (2, 3, 4, 5, 6, 8):
        got = pl.prove_lockstep(jobs[:B], keep=True)
        diff = [j for j in range(B) if got[j] != want[j]]
        print("cap %d B %d:" % (cap, B), "ok" if not diff else "proofs %s differ" % diff)
        if diff:
            bad += 1
            for j in diff[:1]:
                first = next(i for i in range(len(want[j])) if got[j][i] != want[j][i])
                print("   proof %d first differing byte %d of %d" % (j, first, len(want[j])))
    again = {j: pl.prove(j, keep=True) for j in jobs[:2]}
    if any(again[j] != want[j] for j in again):
        bad += 1
        print("cap %d: zk_prove after the batches differs" % cap)
    pl.close()
sys.exit(1 if bad else 0)
