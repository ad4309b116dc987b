#!/bin/bash
# round 6, session 3: does the k = 19 figure depend on the side streams that warm-up and the lone proofs create before the timed region?
cd "$(dirname "$0")/.."
one() {
  local label="$1"; shift
  python bench.py --no-cpu-baseline --k17-steps 0 --steps 40 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-34s value %.2f  repeats %s  single %.2f ms' % ('$label', d['value'], ' '.join('%.1f'%x for x in d.get('value_repeats',[])), d.get('single_proof_ms',0)))
"
}
for rep in 1 2; do
  one "defaults"
  one "all on main (5=2 8=2 9=2)" --opt 5=2 --opt 8=2 --opt 9=2
done
ROW=19,1,1,1,18 python tools/inflight_k17.py 4 2>&1 | grep "proofs/s"
ROW=19,1,1,1,18 OPTS=5=2,8=2,9=2 python tools/inflight_k17.py 4 2>&1 | grep "proofs/s"
