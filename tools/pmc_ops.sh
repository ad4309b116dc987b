# per-kernel SQ / GRBM counters of the hot operators (tools/prof_ops.py): tools/pmc_ops.sh <outdir> [lib tag]
# rocprofv3 counter passes carry --kernel-trace only (no other trace domain), one --pmc set per run.
out=$1; tag=$2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ -n "$tag" ] && export ZKMI355_LIB=$R/webauthn-halo2_amd/build/libzkmi355_$tag.so
mkdir -p $R/$out
timeout 420 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $R/$out/a --output-format csv -- python $R/tools/prof_ops.py 19 3 > $R/$out/a.log 2>&1
timeout 420 rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_SALU -d $R/$out/b --output-format csv -- python $R/tools/prof_ops.py 19 3 > $R/$out/b.log 2>&1
python - <<PY
import csv, glob, collections
for sub in "ab":
    fs = glob.glob("$R/$out/%s/**/*counter_collection.csv" % sub, recursive=True)
    kt = glob.glob("$R/$out/%s/**/*kernel_trace.csv" % sub, recursive=True)
    dur = collections.defaultdict(list)
    for f in kt:
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in fs:
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in sorted(agg.items(), key=lambda kv: -sum(dur.get(kv[0], [0]))):
        d = dur.get(k, [0]); avg = sum(d) / max(len(d), 1)
        if avg < 20000: continue
        print("%-60s launches %3d avg %.1f us  " % (k[-60:], len(d), avg / 1e3) + "  ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(cs.items())))
PY
