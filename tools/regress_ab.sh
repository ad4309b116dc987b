# same box: bench.py under option sets: tools/regress_ab.sh "<opts>" ...   ("-" = none)
for o in "$@"; do
  oo=""; if [ "$o" != "-" ]; then for x in $o; do oo="$oo --opt $x"; done; fi
  python bench.py --no-cpu-baseline --steps 40 $oo 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('opts [$o]', 'value %.1f median %.1f resident %.1f single %.2f accum_ms %.3f exclusive_ms %.3f numa %s'%(d['value'], d['value_median'], d['value_advice_resident'], d['single_proof_ms'], d['roofline']['avg_launch_ms'], d['roofline']['exclusive']['avg_launch_ms'], d['numa_binding']))"
done
