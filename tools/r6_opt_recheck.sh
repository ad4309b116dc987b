#!/bin/bash
# round 6, session 3: bench.py's --opt reached only the FIRST pipeline of a device since the pipelines share an SRS (round 3): the
# option sweeps of rounds 4-6 under four pipelines changed one pipeline in four.  Every one of them again, on all four, one box.
# Usage: gpurun -- 'bash tools/r6_opt_recheck.sh > gpurun_out/r6_opt_recheck.txt 2>&1'
cd "$(dirname "$0")/.."
one() {  # label, extra args
  local label="$1"; shift
  python bench.py --no-cpu-baseline --k17-steps 0 --steps 40 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-34s value %.2f  repeats %s  single %.2f ms' % ('$label', d['value'], ' '.join('%.1f'%x for x in d.get('value_repeats',[])), d.get('single_proof_ms',0)))
"
}
for rep in 1 2; do
  one "defaults"
  one "T1 per bucket (10=1)"          --opt 10=1
  one "columns per pass 1 (2=1)"      --opt 2=1
  one "columns per pass 3 (2=3)"      --opt 2=3
  one "columns per pass 4 (2=4)"      --opt 2=4
  one "NTT radix 2^6 (3=6)"           --opt 3=6
  one "tails on side stream (5=1)"    --opt 5=1
  one "xform side stream (8=1)"       --opt 8=1
  one "msm own stream (9=1)"          --opt 9=1
  one "T1 per bucket + 3 columns"     --opt 10=1 --opt 2=3
done
one "defaults"
