# knobs under the 4-pipeline / one-stream-per-pipeline regime (round 4): pipelines in flight and columns per MSM pass
for cfg in "4 0" "3 0" "6 0" "8 0" "4 2=1" "4 2=3" "4 2=4" "4 3=6" "4 0"; do
  set -- $cfg
  opt=""; [ "$2" != 0 ] && opt="--opt $2"
  python bench.py --no-cpu-baseline --steps 80 --inflight $1 $opt 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('inflight $1 opt $2:', 'proofs/s %.2f (repeats %s) single %.2f h2d %.1f'%(d['value'], ' '.join('%.1f'%x for x in d['value_repeats']), d['single_proof_ms'], d['value_with_h2d']))"
done
