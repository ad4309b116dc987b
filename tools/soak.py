"""Determinism soak: N proofs of the k=19 shape over P concurrent pipelines (default 4: bench.py's regime — tails on the main
streams, double-buffered pass counters; 2: a tail stream each), every proof of a job compared with the first proof of that job
(same witness, same RNG seed -> same bytes).  A race between streams / lanes / pipelines shows up as a differing proof.
usage: soak.py [proofs_per_pipeline] [k] [pipelines] [lockstep]   (lockstep B > 1: every pipeline proves B jobs at a time with
zk_prove_batch, alternating with lone zk_prove calls on the same context — members' workspaces, the shared row stager and the wider
MSM passes must leave nothing behind)"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import webauthn_halo2_amd as zk
from webauthn_halo2_amd import batch, circuit, engine as E

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
k = int(sys.argv[2]) if len(sys.argv) > 2 else 19
npipe = int(sys.argv[3]) if len(sys.argv) > 3 else 4
ls = int(sys.argv[4]) if len(sys.argv) > 4 else 1
p = circuit.K19 if k == 19 else circuit.K17
jobs = list(range(4))
wit = batch.synthesize_jobs(p, jobs)
fixed, copies = batch.structure(p)
OPTS = [tuple(int(x) for x in o.split("=")) for o in os.environ.get("OPTS", "").split(",") if o]  # OPTS=5=2,8=2: zk_ctx_set_option


def factory(dev):
    e = zk.Engine(dev)
    for o, v in OPTS:
        e.set_option(o, v)
    return e


pipes = [batch.Pipeline(0, p, fixed, copies, engine_factory=factory, deterministic_seeds=True)]
for _ in range(npipe - 1):
    pipes.append(batch.Pipeline(0, p, fixed, copies, deterministic_seeds=True, share_srs_with=pipes[0]))
for pl in pipes:
    for j in jobs:
        pl.load(j, wit[j])
ref = {(j, t): pipes[0].prove(j, t, keep=True) for j in jobs for t in (E.ZK_TRANSCRIPT_BLAKE2B, E.ZK_TRANSCRIPT_EVM)}
bad = []


def work_lockstep(pl, off):
    for i in range(0, reps, ls):
        t = E.ZK_TRANSCRIPT_EVM if (i // ls) & 1 else E.ZK_TRANSCRIPT_BLAKE2B
        group = [jobs[(i + off + q) % len(jobs)] for q in range(ls)]
        got = pl.prove_lockstep(group, t, keep=True)
        for q, j in enumerate(group):
            if got[q] != ref[(j, t)]:
                bad.append((off, i, j, t, "lockstep"))
        if (i // ls) % 3 == 2:  # a lone proof in between
            j = jobs[(i + off) % len(jobs)]
            if pl.prove(j, t, keep=True) != ref[(j, t)]:
                bad.append((off, i, j, t, "lone"))


def work(pl, off):
    if ls > 1:
        return work_lockstep(pl, off)
    for i in range(reps):
        j = jobs[(i + off) % len(jobs)]
        t = E.ZK_TRANSCRIPT_EVM if (i // len(jobs)) & 1 else E.ZK_TRANSCRIPT_BLAKE2B
        if pl.prove(j, t, keep=True) != ref[(j, t)]:
            bad.append((off, i, j, t))


t0 = time.time()
ths = [threading.Thread(target=work, args=(pl, q)) for q, pl in enumerate(pipes)]
for t in ths:
    t.start()
for t in ths:
    t.join()
dt = time.time() - t0
print("soak OPTS=%s k=%d lock-step %d: %d proofs over %d pipelines in %.1f s (%.1f proofs/s), mismatches: %d %s" % (os.environ.get("OPTS", "-"), k, ls, npipe * reps, npipe, dt, npipe * reps / dt, len(bad), bad[:5]))
sys.exit(1 if bad else 0)
