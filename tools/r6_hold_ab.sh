#!/bin/bash
# round 6, session 3: a context inside zk_prove counts as active for the whole call (ZK_OPT_ACTIVITY_HOLD: 0 on, 1 = stamps only)
cd "$(dirname "$0")/.."
one() {
  local label="$1"; shift
  python bench.py --no-cpu-baseline --k17-steps 0 --steps 40 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-22s value %.2f  repeats %s  single %.2f ms evm %.2f' % ('$label', d['value'], ' '.join('%.1f'%x for x in d.get('value_repeats',[])), d.get('single_proof_ms',0), d.get('single_proof_evm_ms',0)))
"
}
for rep in 1 2 3; do
  one "stamps only (14=1)" --opt 14=1
  one "held (default)"
done
for rep in 1 2; do
  for o in 14=1 ""; do
    ROW=19,1,1,1,18 OPTS=$o python tools/inflight_k17.py 4 2>&1 | grep "proofs/s"
    OPTS=$o python tools/inflight_k17.py 4 2 2>&1 | grep "proofs/s"
  done
done
