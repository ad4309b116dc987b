#!/usr/bin/env python3
"""One proof's kernel timeline by hardware queue (rocprofv3 --kernel-trace CSV): busy time per queue, the stretches where NO kernel
runs anywhere, and the stretches where only latency-bound reduction-tail kernels run.  usage: timeline2.py <kernel_trace.csv> [from_us to_us]   (the range: every kernel in it, with its queue)"""
import collections, csv, sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "chacha_fr" in r["Kernel_Name"]]
a, b = idx[-4], idx[-3]
seg = rows[a:b]
t0 = int(seg[0]["Start_Timestamp"])
nm = lambda r: r["Kernel_Name"].split("(")[0].replace("zk::", "").replace("void ", "").replace("_kernel", "")
qkey = "Queue_Id" if "Queue_Id" in seg[0] else ("Stream_Id" if "Stream_Id" in seg[0] else None)
print("period ms %.3f kernels %d  (queue column: %s)" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e6, len(seg), qkey))
byq = collections.defaultdict(list)
for r in seg:
    byq[r.get(qkey, "?") if qkey else "?"].append((int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, nm(r)))
for q, iv in sorted(byq.items()):
    busy = sum(e - s for s, e, _ in iv)
    names = collections.Counter(k for _, _, k in iv).most_common(4)
    print("queue %s: %3d kernels, busy %.2f ms: %s" % (q, len(iv), busy / 1e6, ", ".join("%s x%d" % kv for kv in names)))
allk = sorted((int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, nm(r)) for r in seg)
is_tail = lambda k: k.startswith(("msm_wparts", "msm_wrowcol", "msm_wbits"))
# sweep: segments with no kernel, and with tail kernels only
ev = []
for s, e, k in allk:
    ev.append((s, 1, k))
    ev.append((e, -1, k))
ev.sort()
active = collections.Counter()
last = 0
idle = tailonly = 0
gaps = []
for t, d, k in ev:
    if t > last:
        n = sum(active.values())
        if n == 0:
            idle += t - last
            if t - last > 30000:
                gaps.append((last, t - last, "idle"))
        elif all(is_tail(x) for x in active if active[x] > 0):
            tailonly += t - last
            if t - last > 30000:
                gaps.append((last, t - last, "tail only"))
    active[k] += d
    last = t
print("no kernel anywhere: %.2f ms; reduction-tail kernels only: %.2f ms" % (idle / 1e6, tailonly / 1e6))
for at, ln, what in gaps:
    prev = max((x for x in allk if x[1] <= at + 1), key=lambda x: x[1], default=None)
    nxt = min((x for x in allk if x[0] >= at + ln - 1 and not is_tail(x[2])), key=lambda x: x[0], default=None)
    print("  %-9s %5.0f us at %6.0f us: after %s, next %s" % (what, ln / 1e3, at / 1e3, prev[2] if prev else "-", nxt[2] if nxt else "-"))
if len(sys.argv) > 3:
    lo, hi = float(sys.argv[2]) * 1e3, float(sys.argv[3]) * 1e3
    for r in seg:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        if e >= lo and s <= hi:
            print("  q%s %8.0f us +%5.0f us %s" % (r.get(qkey, "?"), s / 1e3, (e - s) / 1e3, nm(r)))
