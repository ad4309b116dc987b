"""Proves exactly N jobs of the k = 19 batch workload on ONE pipeline (bench.py's timed region: every job's advice handed over as a
host buffer, upload inside) and exits — the command tools/pmc_valu.sh runs under rocprofv3's counter collection at two values of
N, so that the per-proof instruction counts are the DIFFERENCE of two runs (set-up and key generation use the same kernels as a
proof and cancel).  One pipeline: a proof's instruction count does not depend on what runs beside it, and counter collection
serialises the dispatches anyway (four pipelines x 48 proofs did not finish in 15 minutes).   usage: valu_proofs.py N"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

n = int(sys.argv[1])
bench.SYNTH_PROCESSES = 1  # no process pool under rocprofv3 (its signal handler in the pool's workers stalls the run)
wl = bench.ProofWorkload(0, 0, 1, 1, n, 1)
wl.run_with_h2d(wl.jobs[:n])
assert len(wl.proofs) == n and bench.check_against_oracle_digests(wl.proofs) == min(n, 256)
print("proved", n)
wl.close()
