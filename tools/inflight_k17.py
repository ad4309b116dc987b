"""Proofs/s of the proving server's default shape (k = 17, EVM transcript + GWC) against the number of pipelines in flight on
one GPU (bench.py measures k = 19, where two are best).  usage: inflight_k17.py [pipelines ...]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from webauthn_halo2_amd import batch, circuit, engine as E  # noqa: E402

p = circuit.K17
jobs = list(range(8))
wit = batch.synthesize_jobs(p, jobs)
fixed, copies = batch.structure(p)
for npipe in [int(x) for x in (sys.argv[1:] or ["1", "2", "3", "4"])]:
    pipes = [batch.Pipeline(0, p, fixed, copies, deterministic_seeds=True)]
    for _ in range(npipe - 1):
        pipes.append(batch.Pipeline(0, p, fixed, copies, deterministic_seeds=True, share_srs_with=pipes[0]))
    for pl in pipes:
        for j in jobs:
            pl.load(j, wit[j])
        pl.prove(0, E.ZK_TRANSCRIPT_EVM, keep=True)
    reps = 60

    def work(pl):
        for i in range(reps):
            pl.prove(jobs[i % len(jobs)], E.ZK_TRANSCRIPT_EVM, keep=True)

    ths = [threading.Thread(target=work, args=(pl,)) for pl in pipes]
    t0 = time.perf_counter()
    [t.start() for t in ths]
    [t.join() for t in ths]
    dt = time.perf_counter() - t0
    print(f"k=17 EVM, {npipe} pipelines: {npipe * reps / dt:6.1f} proofs/s ({dt / reps * 1e3:.2f} ms per proof and pipeline)", flush=True)
    for pl in pipes[::-1]:
        pl.close()
