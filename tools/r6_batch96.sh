#!/bin/bash
cd "$(dirname "$0")/.."
timeout 400 python -m pytest tests/test_gpu_cosets3.py tests/test_gpu_ops.py -x -q 2>&1 | tail -2
for rep in 1 2; do
  for o in 13=1 13=2; do
    echo "bench_rows OPTS=$o"
    OPTS=$o ROWS=16,15,14,13,12,11 python tools/bench_rows.py 7 2>&1 | grep -v "^degree"
  done
done
