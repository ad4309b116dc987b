cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for v in "$@"; do
  if [ "$v" = base ]; then unset ZKMI355_LIB; else export ZKMI355_LIB=$R/webauthn-halo2_amd/build/libzkmi355_$v.so; fi
  rm -rf /tmp/sc_$v; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sc_$v -- python $R/tools/prof_ops.py 19 4 > /dev/null 2>&1
  f=$(find /tmp/sc_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v"; python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    n=r['Name'].split('(')[0]
    if any(k in n for k in ('scatter','recode','msm_scan','accumulate','gather','bitsum')): print('%-40s calls %4s avg %9.1f us' % (n[-40:], r['Calls'], float(r['AverageNs'])/1e3))
"
done
