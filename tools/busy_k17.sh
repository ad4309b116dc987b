# GPU busy fraction (union of kernel intervals) while k=17 proofs run over N pipelines: tools/busy_k17.sh <N>
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/busy_k17; mkdir -p $O
( cd $R && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/raw -- python tools/inflight_k17.py $1 > $O/run.log 2>&1 )
f=$(find $O/raw -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
t1 = rows[-1][1]
t0 = t1 - int(0.25e9)            # the last 250 ms: steady state
seg = [(max(a, t0), b) for a, b, _ in rows if b > t0]
busy, cur_a, cur_b = 0, None, None
for a, b in seg:
    if cur_b is None or a > cur_b:
        if cur_b is not None: busy += cur_b - cur_a
        cur_a, cur_b = a, b
    else:
        cur_b = max(cur_b, b)
busy += cur_b - cur_a
print("kernels in window", len(seg), "busy fraction %.3f" % (busy / (t1 - t0)), "kernel-seconds per second %.2f" % (sum(b - a for a, b in seg) / (t1 - t0)))
PY
grep "pipelines" $O/run.log
rm -rf $O/raw
