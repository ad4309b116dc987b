#!/bin/bash
# round 6, session 3: the k = 17 shape's knobs under four pipelines (its tails are a quarter of its kernel time): window bits, T1 form,
# columns per pass.  Usage: gpurun -- 'bash tools/r6_k17_knobs.sh > gpurun_out/r6_k17_knobs.txt 2>&1'
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for o in "" 1=15 1=14 10=1 2=3 2=4 2=8 6=8; do
    OPTS=$o python tools/inflight_k17.py 1 4 2>&1 | grep "proofs/s"
  done
done
