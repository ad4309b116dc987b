# HIP API calls of lone proofs (host-side launch path): tools/hip_api_stats.sh [K19|K17] [blake2b|evm]
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/hipapi; mkdir -p $O
( cd $R && timeout 600 rocprofv3 --hip-trace --stats --output-format csv -d $O/raw -- python tools/trace_one.py ${1:-K19} ${2:-blake2b} 20 > $O/run.log 2>&1 )
f=$(find $O/raw -name "*hip_api_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print("HIP API, calls, total ms, avg us   (20 proofs + set-up)")
for r in rows[:14]:
    print(f"{r['Name']:36s} {r['Calls']:>8s} {float(r['TotalDurationNs'])/1e6:10.2f} {float(r['AverageNs'])/1e3:9.2f}")
PY
grep bytes $O/run.log
rm -rf $O/raw
