#!/bin/bash
# round 6, session 3: where the reduction tails should run under four pipelines, by column length: ZK_OPT_MSM_TAIL_STREAM 1 (side) / 2 (main)
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for row in 18,2,1,1,17 16,8,2,1,15; do
    for o in 5=1 5=2; do
      ROW=$row OPTS=$o python tools/inflight_k17.py 4 2>&1 | grep "proofs/s"
    done
  done
done
