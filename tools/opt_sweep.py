"""Engine options (zk_ctx_set_option) against single-proof time and two-pipeline throughput at k=19 (or K=17/18):
usage: opt_sweep.py <option id> <values...>   (1 = MSM window, 2 = MSM columns per pass, 3 = NTT max radix log2)"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import webauthn_halo2_amd as zk
from webauthn_halo2_amd import batch, circuit, engine as E

K = os.environ.get("K", "19")
p = circuit.K17 if K == "17" else circuit.K19
opt = int(sys.argv[1])
jobs = list(range(4))
wit = batch.synthesize_jobs(p, jobs)
fixed, copies = batch.structure(p)


def mk(v):
    def factory(dev):
        e = zk.Engine(dev)
        e.set_option(opt, v)
        return e
    return factory


for v in [int(x) for x in sys.argv[2:]]:
    pipes = [batch.Pipeline(0, p, fixed, copies, engine_factory=mk(v), deterministic_seeds=True) for _ in range(2)]
    for pl in pipes:
        for j in jobs:
            pl.load(j, wit[j])
    for pl in pipes:
        pl.prove(0, keep=True)
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); pipes[0].prove(1, keep=True); ts.append((time.perf_counter() - t0) * 1e3)
    reps = 40

    def work(pl):
        for i in range(reps):
            pl.prove(jobs[i % 4], keep=True)
    t0 = time.perf_counter()
    ths = [threading.Thread(target=work, args=(pl,)) for pl in pipes]
    [t.start() for t in ths]; [t.join() for t in ths]
    dt = time.perf_counter() - t0
    print("option %d = %d: single %.2f ms  two pipelines %.1f proofs/s" % (opt, v, min(ts), 2 * reps / dt), flush=True)
    for pl in pipes:
        pl.close()
