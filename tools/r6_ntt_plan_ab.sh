#!/bin/bash
# round 6, session 3: two-pass NTT plans on larger tiles for 2^15 .. 2^20 (the default since) against three passes of 2^7 on the 2^9
# tile (ZK_OPT_NTT_MAX_RADIX_LOG2 = 7: the plan of rounds 3-5): k = 17 EVM over 1 / 4 pipelines, bench.py k = 19, the config rows
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for o in 3=7 ""; do
    OPTS=$o python tools/inflight_k17.py 1 4 2>&1 | grep "proofs/s"
  done
done
one() {
  local label="$1"; shift
  python bench.py --no-cpu-baseline --k17-steps 0 --steps 40 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-26s value %.2f  repeats %s  single %.2f ms evm %.2f' % ('$label', d['value'], ' '.join('%.1f'%x for x in d.get('value_repeats',[])), d.get('single_proof_ms',0), d.get('single_proof_evm_ms',0)))
"
}
for rep in 1 2 3; do
  one "three passes of 2^7 (3=7)" --opt 3=7
  one "plan by size (default)"
done
for o in 3=7 ""; do
  echo "bench_rows OPTS=$o"
  OPTS=$o python tools/bench_rows.py 7 2>&1 | grep -v "^degree"
done
