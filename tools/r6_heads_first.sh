#!/bin/bash
# round 6, session 3: heads first (ZK_OPT_MSM_HEADS_FIRST: 0 on, 1 off) - lone proofs (tools/single_ab.py) and bench.py
cd "$(dirname "$0")/.."
timeout 600 python -m pytest tests/test_gpu_prover.py tests/test_gpu_fullsize_parity.py tests/test_gpu_audit.py -x -q 2>&1 | tail -3
for rep in 1 2 3; do
  for o in 15=1 15=0; do
    OPTS=$o python tools/single_ab.py 2>&1 | tail -2
  done
done
one() {
  local label="$1"; shift
  python bench.py --no-cpu-baseline --k17-steps 0 --steps 40 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-22s value %.2f  repeats %s  single %.2f ms evm %.2f' % ('$label', d['value'], ' '.join('%.1f'%x for x in d.get('value_repeats',[])), d.get('single_proof_ms',0), d.get('single_proof_evm_ms',0)))
"
}
for rep in 1 2; do
  one "one after the other" --opt 15=1
  one "heads first"
done
