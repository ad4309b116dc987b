"""Proofs/s of the proving server's default shape (k = 17, EVM transcript + GWC) against the number of pipelines in flight on
one GPU (bench.py measures k = 19).  usage: inflight_k17.py [pipelines[xlockstep] ...]   ("2x4": two pipelines, each proving four
jobs at a time in lock-step, zk_prove_batch)"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from webauthn_halo2_amd import batch, circuit, engine as E  # noqa: E402

p = circuit.K17
if os.environ.get("ROW"):  # another bench_ecdsa.config row: ROW=18,2,1,1,17 (degree, advice, lookup advice, fixed, lookup bits)
    v = [int(x) for x in os.environ["ROW"].split(",")]
    p = circuit.CircuitParams(degree=v[0], num_advice=v[1], num_lookup_advice=v[2], num_fixed=v[3], lookup_bits=v[4])
jobs = list(range(8))
wit = batch.synthesize_jobs(p, jobs)
fixed, copies = batch.structure(p)
OPTS = [tuple(int(x) for x in o.split("=")) for o in os.environ.get("OPTS", "").split(",") if o]  # OPTS=5=1,2=4: zk_ctx_set_option


def configure(e):
    for o, v in OPTS:
        e.set_option(o, v)
    return e


def factory(dev):
    return configure(E.Engine(dev))


factory.configure = configure  # (batch.Pipeline: the options also reach the pipelines that share the first one's SRS)


for spec in (sys.argv[1:] or ["1", "2", "3", "4"]):
    npipe, ls = (int(x) for x in (spec.split("x") + ["1"])[:2])
    pipes = [batch.Pipeline(0, p, fixed, copies, engine_factory=factory, deterministic_seeds=True)]
    inter = bool(os.environ.get("INTERLEAVE"))  # a lone proof on every pipeline BEFORE the next one is created (side streams made early)
    if inter:
        pipes[0].load(0, wit[0])
        pipes[0].prove(0, E.ZK_TRANSCRIPT_EVM, keep=True)
    for _ in range(npipe - 1):
        pipes.append(batch.Pipeline(0, p, fixed, copies, engine_factory=factory, deterministic_seeds=True, share_srs_with=pipes[0]))
        if inter:
            pipes[-1].load(0, wit[0])
            pipes[-1].prove(0, E.ZK_TRANSCRIPT_EVM, keep=True)
    for pl in pipes:
        for j in jobs:
            if not (inter and j == 0):
                pl.load(j, wit[j])
        pl.prove(0, E.ZK_TRANSCRIPT_EVM, keep=True)
    reps = 64

    def work(pl):
        if ls > 1:
            for i in range(0, reps, ls):
                pl.prove_lockstep([jobs[(i + q) % len(jobs)] for q in range(ls)], E.ZK_TRANSCRIPT_EVM, keep=True)
            return
        for i in range(reps):
            pl.prove(jobs[i % len(jobs)], E.ZK_TRANSCRIPT_EVM, keep=True)

    ths = [threading.Thread(target=work, args=(pl,)) for pl in pipes]
    t0 = time.perf_counter()
    [t.start() for t in ths]
    [t.join() for t in ths]
    dt = time.perf_counter() - t0
    print(f"k={p.degree} EVM OPTS={os.environ.get('OPTS', '-')}, {npipe} pipelines x lock-step {ls}: {npipe * reps / dt:6.1f} proofs/s ({dt / reps * 1e3:.2f} ms per proof and pipeline)", flush=True)
    for pl in pipes[::-1]:
        pl.close()
