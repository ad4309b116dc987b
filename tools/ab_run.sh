#!/bin/bash
# A/B harness: runs bench.py under each "NAME=VALUE" environment setting given as arguments (or none)
run() {
  for fl in 1 2; do
    python bench.py --inflight $fl --no-cpu-baseline --steps 12 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', 'inflight', $fl, 'proofs/s %.2f single %.2f accum_ms %.3f'%(d['value'], d['single_proof_ms'], d['roofline']['avg_launch_ms']))"
  done
}
run base
for kv in "$@"; do
  env "$kv" bash -c "$(declare -f run); run '$kv'"
done
