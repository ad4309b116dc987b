# VALU instructions of a k = 19 proof by kernel (roofline.valu_issue of bench.py): tools/pmc_valu.sh <tag>
# two counter passes of tools/valu_proofs.py (8 and 40 proofs), --kernel-trace only beside --pmc, one counter set per run;
# -> profiles/<tag>_proof_k19_pmc_valu.csv (per kernel: launches and counters PER PROOF = (run40 - run8) / 32)
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/${tag}_pmc_valu
mkdir -p $O
for n in 8 40; do
  timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES -d $O/n$n --output-format csv -- python $R/tools/valu_proofs.py $n > $O/n$n.log 2>&1
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
    return agg, cnt
a8, c8 = load(8)
a40, c40 = load(40)
rows = []
for k in a40:
    per = {c: (a40[k][c] - a8.get(k, {}).get(c, 0.0)) / 32.0 for c in a40[k]}
    launches = (c40[k] - c8.get(k, 0)) / 32.0
    if launches <= 0 or per.get("SQ_INSTS_VALU", 0) <= 0:
        continue
    rows.append([k, "%.3f" % launches, "%.0f" % per.get("SQ_INSTS_VALU", 0), "%.0f" % per.get("SQ_BUSY_CYCLES", 0), "%.0f" % per.get("GRBM_GUI_ACTIVE", 0), "%.0f" % per.get("SQ_WAVES", 0)])
rows.sort(key=lambda r: -float(r[2]))
with open("profiles/%s_proof_k19_pmc_valu.csv" % tag, "w", newline="") as fh:
    w = csv.writer(fh)
    w.writerow(["kernel", "launches_per_proof", "SQ_INSTS_VALU_per_proof", "SQ_BUSY_CYCLES_per_proof", "GRBM_GUI_ACTIVE_per_proof", "SQ_WAVES_per_proof"])
    w.writerows(rows)
tot = sum(float(r[2]) for r in rows)
print("VALU wave-instructions per k = 19 proof: %.4g over %d kernels" % (tot, len(rows)))
for r in rows[:14]:
    print("  %-58s %7s launches  %.4g  (%.1f %%)" % (r[0][-58:], r[1], float(r[2]), 100 * float(r[2]) / tot))
PY
mkdir -p gpurun_out/${tag}_profiles && cp profiles/${tag}_proof_k19_pmc_valu.csv gpurun_out/${tag}_profiles/
rm -rf $O/n8 $O/n40
