#!/bin/bash
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
  for o in 5=1 5=2 6=8 ""; do
    OPTS=$o python tools/inflight_k17.py 4 2>&1 | grep "proofs/s"
  done
done
