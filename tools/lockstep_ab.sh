# lock-step regimes under bench.py on ONE box: tools/lockstep_ab.sh "<inflight>x<lockstep>[:opt=val,..]" ...   (4x1 = round 4's regime)
for v in "$@"; do
  reg=${v%%:*}; opts=""; if [ "$reg" != "$v" ]; then for o in $(echo ${v#*:} | tr ',' ' '); do opts="$opts --opt $o"; done; fi
  inf=${reg%x*}; ls=${reg#*x}
  python bench.py --no-cpu-baseline --steps ${STEPS:-48} --inflight $inf --lockstep $ls $opts 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', 'proofs/s %.2f (repeats %s) single %.2f with_h2d %.2f accum_ms %.3f cols/launch %.2f'%(d['value'], ' '.join('%.1f'%x for x in d['value_repeats']), d['single_proof_ms'], d.get('value_with_h2d',0), d['roofline']['avg_launch_ms'], d['roofline']['columns_per_launch']))"
done
