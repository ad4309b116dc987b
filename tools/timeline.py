#!/usr/bin/env python3
"""One proof's kernel timeline from a rocprofv3 --kernel-trace CSV: per-kernel totals, union busy
time, and the stretches where the GPU idles or only a side-stream (MSM tail) kernel runs.
usage: timeline.py <kernel_trace.csv> [--full]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "chacha_fr" in r["Kernel_Name"]]
a, b = idx[-4], idx[-3]
seg = rows[a:b]
t0 = int(seg[0]["Start_Timestamp"])


def nm(r):
    return r["Kernel_Name"].split("(")[0].replace("zk::", "").replace("void ", "").replace("_kernel", "")


print("period ms", (int(rows[b]["Start_Timestamp"]) - t0) / 1e6, "kernels", len(seg))
agg = collections.defaultdict(lambda: [0, 0])
for r in seg:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg[nm(r)][0] += d
    agg[nm(r)][1] += 1
tot = 0
for k, (d, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:18]:
    print(f"{k:32s} {c:4d} {d / 1e6:8.3f} ms")
for k, (d, c) in agg.items():
    tot += d
print("sum", tot / 1e6)
def is_tail(k):
    return k.startswith("msm_gather") or k.startswith("msm_bitsum") or k.startswith("msm_wparts") or k.startswith("msm_wrowcol") or k.startswith("msm_wbits")


main = [(int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, nm(r)) for r in seg if not is_tail(nm(r))]
allk = [(int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, nm(r)) for r in seg]


def union(iv):
    busy = 0
    cs, ce = iv[0][0], iv[0][1]
    for s, e, _ in iv[1:]:
        if s > ce:
            busy += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return busy + ce - cs


print("union busy (all) ms", union(allk) / 1e6, " union busy (non-tail) ms", union(main) / 1e6)
# gaps in the non-tail timeline
last = main[0][1]
prev = main[0][2]
for s, e, k in main[1:]:
    if s - last > 50000:
        print(f"  main-stream gap {(s - last) / 1e3:6.0f} us at {last / 1e3:8.0f}: after {prev}, before {k}")
    if e > last:
        last, prev = e, k
if "--full" in sys.argv:
    for s, e, k in allk:
        print(f"{s / 1e3:9.0f} {(e - s) / 1e3:7.0f}us {k}")
