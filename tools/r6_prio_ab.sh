#!/bin/bash
# round 6, session 3: pipelines of one device at DIFFERENT stream priorities (ZK_OPT_STREAM_PRIORITY = 12; 0 normal, 1 high, 2 low)
# against four equal ones, alternating on one box; and how the figure depends on the number of timed steps (fill / drain of the
# four pipelines).  Usage: gpurun -- 'bash tools/r6_prio_ab.sh > gpurun_out/r6_prio_ab.txt 2>&1'
cd "$(dirname "$0")/.."
one() {  # label, extra args
  local label="$1"; shift
  python bench.py --no-cpu-baseline --k17-steps 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-28s value %.2f  repeats %s  single %.2f ms' % ('$label', d['value'], ' '.join('%.1f'%x for x in d.get('value_repeats',[])), d.get('single_proof_ms',0)))
"
}
for rep in 1 2; do
  one "equal"            --steps 40
  one "prio 1,0,0,2"     --steps 40 --opt 12=1,0,0,2
  one "prio 1,1,2,2"     --steps 40 --opt 12=1,1,2,2
  one "prio 1,2,2,2"     --steps 40 --opt 12=1,2,2,2
  one "prio 1,0,2,0"     --steps 40 --opt 12=1,0,2,0
done
one "equal steps 20"   --steps 20 --warmup 5
one "equal steps 100"  --steps 100
one "equal steps 200"  --steps 200
