# instruction-cache counters of the hot kernels, alone (tools/prof_ops.py) and under four overlapping pipelines (bench.py)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/icache; mkdir -p $O
rocprofv3 --list-avail 2>/dev/null | grep -i -E "ICACHE|SQC_INST" | head -20 > $O/avail.txt
timeout 420 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/solo --output-format csv -- python $R/tools/prof_ops.py 19 3 > $O/solo.log 2>&1
( cd $R && timeout 420 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/load --output-format csv -- python bench.py --no-cpu-baseline --steps 40 > $O/load.log 2>&1 )
python3 - <<PY
import csv, glob, collections
for sub in ("solo", "load"):
    fs = glob.glob("$O/%s/**/*counter_collection.csv" % sub, recursive=True)
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in fs:
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0].replace("zk::","").replace("void ","")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", sub)
    for k, cs in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("SQ_WAVE_CYCLES",[0])))[:12]:
        g = lambda n: sum(cs.get(n,[0]))/max(len(cs.get(n,[1])),1)
        req, miss = g("SQC_ICACHE_REQ"), g("SQC_ICACHE_MISSES")
        print("  %-34s launches %4d  icache req %.3g  misses %.3g  (%.2f %%)  wave cycles %.3g" % (k[:34], len(cs.get("SQC_ICACHE_REQ",[])), req, miss, 100*miss/max(req,1), g("SQ_WAVE_CYCLES")))
PY
cat $O/avail.txt | head -8
rm -rf $O/solo $O/load
