#!/bin/bash
# round 6, session 3: eight hardware queues with DELIBERATE placement (four main streams + four tail streams on a queue each)
cd "$(dirname "$0")/.."
one() {
  local label="$1"; shift
  python bench.py --no-cpu-baseline --k17-steps 0 --steps 40 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-40s value %.2f  repeats %s  single %.2f ms' % ('$label', d['value'], ' '.join('%.1f'%x for x in d.get('value_repeats',[])), d.get('single_proof_ms',0)))
"
}
for rep in 1 2; do
  one "4 queues, defaults"
  GPU_MAX_HW_QUEUES=8 one "8 queues, defaults (tails on main)"
  GPU_MAX_HW_QUEUES=8 one "8 queues, tails on their own queues" --opt 5=1
  GPU_MAX_HW_QUEUES=8 one "8 queues, 6 pipelines, tails on main" --inflight 6
done
