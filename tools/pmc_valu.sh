# VALU instructions of a k = 19 proof by kernel (roofline.valu_issue of bench.py): tools/pmc_valu.sh <tag>
# two counter passes of tools/valu_proofs.py (1 and 5 proofs on one pipeline), --kernel-trace only beside --pmc, one counter set per run;
# -> profiles/<tag>_proof_k19_pmc_valu.csv (per kernel: launches and counters PER PROOF = (run5 - run1) / 4)
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/${tag}_pmc_valu
mkdir -p $O
for n in 1 5; do
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES -d $O/n$n --output-format csv -- python $R/tools/valu_proofs.py $n > $O/n$n.log 2>&1
  tail -1 $O/n$n.log
done
cd $R
python - "$O" "$tag" <<'PY'
import collections, csv, glob, sys
O, tag = sys.argv[1], sys.argv[2]
def load(n):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.Counter()
    for f in glob.glob("%s/n%d/**/*counter_collection.csv" % (O, n), recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (r.get("Dispatch_Id"), k)
            if key not in seen:
                seen.add(key)
                cnt[k] += 1
    for f in glob.glob("%s/n%d/**/*kernel_trace.csv" % (O, n), recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"].split("(")[0]]["duration_ns"] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    return agg, cnt
a8, c8 = load(1)
a40, c40 = load(5)
rows = []
for k in a40:
    per = {c: (a40[k][c] - a8.get(k, {}).get(c, 0.0)) / 4.0 for c in a40[k]}
    launches = (c40[k] - c8.get(k, 0)) / 4.0
    if launches <= 0 or per.get("SQ_INSTS_VALU", 0) <= 0:
        continue
    rows.append([k, "%.3f" % launches, "%.0f" % per.get("SQ_INSTS_VALU", 0), "%.0f" % per.get("GRBM_GUI_ACTIVE", 0), "%.0f" % per.get("SQ_WAVES", 0), "%.0f" % per.get("duration_ns", 0)])
rows.sort(key=lambda r: -float(r[2]))
with open("profiles/%s_proof_k19_pmc_valu.csv" % tag, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "launches_per_proof", "SQ_INSTS_VALU_per_proof", "GRBM_GUI_ACTIVE_per_proof", "SQ_WAVES_per_proof", "duration_ns_per_proof_under_the_profiler"])
    w.writerows(rows)
tot = sum(float(r[2]) for r in rows)
print("VALU wave-instructions per k = 19 proof: %.4g over %d kernels" % (tot, len(rows)))
for r in rows[:14]:
    clk = float(r[3]) / 8.0 / max(float(r[5]), 1.0)  # GRBM_GUI_ACTIVE is summed over the 8 XCDs
    print("  %-58s %7s launches  %.4g  (%.1f %%)  GRBM clock %.2f GHz" % (r[0][-58:], r[1], float(r[2]), 100 * float(r[2]) / tot, clk))
PY
mkdir -p gpurun_out/${tag}_profiles && cp profiles/${tag}_proof_k19_pmc_valu.csv gpurun_out/${tag}_profiles/
rm -rf $O/n1 $O/n5
