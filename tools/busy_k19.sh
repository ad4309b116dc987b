# GPU busy fraction and overlap while bench.py proves (k = 19, N pipelines): tools/busy_k19.sh <N>
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/busy_k19; mkdir -p $O
( cd $R && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/raw -- python bench.py --no-cpu-baseline --k17-steps 0 --steps 120 --inflight $1 > $O/run.log 2>&1 )
f=$(find $O/raw -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("zk::", "").replace("void ", "")) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# the densest 1.0 s window = the timed batches (3 repeats of 120 proofs)
quot = [a for a, b, k in rows if k.startswith("quotient_kernel")]
t1 = quot[-40]; t0 = quot[-340]   # 300 proofs of steady state
seg = [(max(a, t0), min(b, t1), k) for a, b, k in rows if b > t0 and a < t1]
busy, cur_a, cur_b = 0, None, None
for a, b, _ in seg:
    if cur_b is None or a > cur_b:
        if cur_b is not None: busy += cur_b - cur_a
        cur_a, cur_b = a, b
    else:
        cur_b = max(cur_b, b)
busy += cur_b - cur_a
W = t1 - t0
print("window %.1f ms, 300 proofs -> %.2f ms/proof; busy fraction %.3f; kernel-seconds per second %.2f" % (W / 1e6, W / 300 / 1e6, busy / W, sum(b - a for a, b, _ in seg) / W))
agg = collections.defaultdict(lambda: [0, 0])
for a, b, k in seg:
    agg[k][0] += b - a; agg[k][1] += 1
for k, (d, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:16]:
    print("  %-34s %6.1f/proof  avg %8.1f us  %6.3f ms/proof" % (k[:34], c / 300, d / c / 1e3, d / 300 / 1e6))
# time with exactly j of the big kernels (accumulate / ntt / quotient) running
ev = []
for a, b, k in seg:
    if k.startswith("msm_wacc_fast") or k.startswith("ntt_pass") or k.startswith("quotient_kernel"):
        ev.append((a, 1)); ev.append((b, -1))
ev.sort()
lvl, last, hist = 0, t0, collections.Counter()
for t, d in ev:
    hist[lvl] += t - last; last = t; lvl += d
print("  time share with j chip-filling kernels in flight:", "  ".join("%d: %.2f" % (j, hist[j] / W) for j in sorted(hist)))
PY
tail -1 $O/run.log | cut -c1-120
rm -rf $O/raw
