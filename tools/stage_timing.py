#!/usr/bin/env python3
"""Where the time goes when a loader thread stages job i+1 while the pipeline proves job i (k=19): per-job stage / wait /
adopt / prove times, for 1 and 2 pipelines.  Usage: stage_timing.py [jobs]"""
import os
import queue
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import webauthn_halo2_amd as zk  # noqa: E402
from webauthn_halo2_amd import batch  # noqa: E402


def main():
    njobs = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    k = 19
    p = zk.circuit.CircuitParams(degree=k, num_advice=1, num_lookup_advice=1, num_fixed=1, lookup_bits=18)
    asg = zk.circuit.synthesize(p, 0x5EED0019)
    fixed = np.stack([asg.to_limbs(c) for c in asg.fixed])
    wit = [asg.to_limbs(c) for c in asg.advice]
    first = batch.Pipeline(0, p, fixed, asg.copies, deterministic_seeds=True)
    pipes = [first, batch.Pipeline(0, p, fixed, asg.copies, deterministic_seeds=True, share_srs_with=first)]
    for pl in pipes:
        pl.load(0, wit)
        pl.prove(0)
    for npipe in (1, 2):
        for mode in ("inline", "staged"):
            rec = {q: [] for q in range(npipe)}

            def work(q):
                pl = pipes[q]
                if mode == "inline":
                    for j in range(njobs):
                        t0 = time.perf_counter()
                        pl.load(j, wit)
                        t1 = time.perf_counter()
                        pl.prove(j)
                        rec[q].append((t1 - t0, 0, 0, time.perf_counter() - t1))
                    return
                st = queue.Queue(maxsize=2)
                stage_t = []

                def load():
                    for j in range(njobs):
                        t0 = time.perf_counter()
                        s = pl.stage(wit)
                        stage_t.append(time.perf_counter() - t0)
                        st.put((j, s))

                lt = threading.Thread(target=load)
                lt.start()
                for _ in range(njobs):
                    t0 = time.perf_counter()
                    j, s = st.get()
                    t1 = time.perf_counter()
                    pl.adopt(j, s)
                    t2 = time.perf_counter()
                    pl.prove(j)
                    rec[q].append((stage_t[j], t1 - t0, t2 - t1, time.perf_counter() - t2))
                lt.join()

            t0 = time.perf_counter()
            ths = [threading.Thread(target=work, args=(q,)) for q in range(npipe)]
            [t.start() for t in ths]
            [t.join() for t in ths]
            dt = time.perf_counter() - t0
            a = np.array(sum(rec.values(), [])) * 1e3
            print(f"pipes={npipe} {mode:7s}: {npipe * njobs / dt:6.1f} proofs/s; mean ms load/stage {a[:, 0].mean():.2f} "
                  f"wait {a[:, 1].mean():.2f} adopt {a[:, 2].mean():.2f} prove {a[:, 3].mean():.2f} "
                  f"(max stage {a[:, 0].max():.2f}, max prove {a[:, 3].max():.2f})", flush=True)
    for pl in pipes[::-1]:
        pl.close()


if __name__ == "__main__":
    main()
