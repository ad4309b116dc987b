# Kernel-time profile of one command on the GPU box: tools/prof_cmd.sh <tag> <command...>
# -> gpurun_out/<tag>/kernel_stats.csv (rocprofv3 --kernel-trace --stats), top rows printed
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/$tag
mkdir -p $O
( cd $R && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -- "$@" > $O/run.log 2>&1 )
f=$(find $O/raw -name "*kernel_stats.csv" | head -1)
cp $f $O/kernel_stats.csv
rm -rf $O/raw
python3 - "$O/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:28]:
    print("%-60s calls %6s  avg %10.1f us  total %8.2f ms  %5s%%" % (r["Name"].split("(")[0][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
tail -5 $O/run.log
