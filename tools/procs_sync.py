"""Whole-GPU proofs/s of P worker PROCESSES x N pipelines (x lock-step B) on one device, measured over ONE common window: the
workers meet at a barrier after their own set-up and warm-up, prove for T seconds, and the proofs COMPLETED inside the window are
summed (tools/procs_ab.sh adds up the processes' own clocks, which need not overlap).
usage: procs_sync.py k P N [B] [seconds] [opt=val,..]      e.g. procs_sync.py 19 2 2"""
import multiprocessing as mp
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(idx, k, N, B, secs, opts, bar, counts, t_start):
    from webauthn_halo2_amd import batch, circuit, engine as E

    p = circuit.K19 if k == 19 else circuit.K17
    tk = E.ZK_TRANSCRIPT_BLAKE2B if k == 19 else E.ZK_TRANSCRIPT_EVM
    jobs = list(range(8))
    wit = batch.synthesize_jobs(p, jobs, processes=4)
    fixed, copies = batch.structure(p)

    def factory(dev):
        e = E.Engine(dev)
        for o, v in opts:
            e.set_option(o, v)
        return e

    pipes = [batch.Pipeline(0, p, fixed, copies, engine_factory=factory, deterministic_seeds=True)]
    for _ in range(N - 1):
        pipes.append(batch.Pipeline(0, p, fixed, copies, deterministic_seeds=True, share_srs_with=pipes[0]))
    for pl in pipes:
        for j in jobs:
            pl.load(j, wit[j])
        if B > 1:
            pl.prove_lockstep(jobs[:B], tk, keep=True)
        pl.prove(0, tk, keep=True)
    done = [0] * N
    stop = [False]

    def work(q):
        i = 0
        while not stop[0]:
            if B > 1:
                pipes[q].prove_lockstep([jobs[(i + t) % 8] for t in range(B)], tk, keep=True)
            else:
                pipes[q].prove(jobs[i % 8], tk, keep=True)
            i += B
            if time.time() >= t_start.value + secs:
                break
            done[q] += B
    bar.wait()
    if idx == 0:
        t_start.value = time.time() + 0.05
    bar.wait()
    while time.time() < t_start.value:
        pass
    ths = [threading.Thread(target=work, args=(q,)) for q in range(N)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    counts[idx] = sum(done)
    for pl in pipes[::-1]:
        pl.close()


if __name__ == "__main__":
    k, P, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    B = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    secs = float(sys.argv[5]) if len(sys.argv) > 5 else 4.0
    opts = [tuple(int(x) for x in o.split("=")) for o in (sys.argv[6].split(",") if len(sys.argv) > 6 and sys.argv[6] else [])]
    ctx = mp.get_context("spawn")
    bar = ctx.Barrier(P)
    counts = ctx.Array("i", P)
    t_start = ctx.Value("d", 0.0)
    ps = [ctx.Process(target=worker, args=(i, k, N, B, secs, opts, bar, counts, t_start)) for i in range(P)]
    [p.start() for p in ps]
    [p.join() for p in ps]
    tot = sum(counts[:])
    print("k=%d  %d process(es) x %d pipelines x lock-step %d%s: %s proofs completed in a common %.1f s window = %.1f proofs/s"
          % (k, P, N, B, (" opts " + sys.argv[6]) if opts else "", " + ".join(str(c) for c in counts[:]), secs, tot / secs), flush=True)
