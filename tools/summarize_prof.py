"""Condenses rocprofv3 outputs under gpurun_out/ into small tracked files under profiles/.
usage: summarize_prof.py <tag> <stats_dir> [<pmc_fetch_dir> <pmc_write_dir>]"""
import collections, csv, glob, os, shutil, sys

tag, stats = sys.argv[1], sys.argv[2]
os.makedirs("profiles", exist_ok=True)
src = glob.glob(os.path.join(stats, "*kernel_stats.csv"))[0]
shutil.copy(src, f"profiles/{tag}_kernel_stats.csv")
if len(sys.argv) > 3:
    out = [["kernel", "counter", "launches", "avg_value_per_launch", "unit_note"]]
    for d in sys.argv[3:]:
        f = glob.glob(os.path.join(d, "*counter_collection.csv"))[0]
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            agg[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            out.append([k, c, len(v), "%.1f" % (sum(v) / len(v)), "KiB as reported by rocprofv3 (FETCH_SIZE under-reports wide coalesced reads 2x on gfx950, see MI355X_MICROARCH.md HBM)"])
    csv.writer(open(f"profiles/{tag}_pmc_hbm.csv", "w")).writerows(out)
print("wrote profiles/%s_*" % tag)
