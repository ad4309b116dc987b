"""profiles/ must describe the code that is committed: every engine kernel named in the newest round's rocprofv3 kernel
statistics exists in csrc/ (round 4's summaries listed a kernel that a later commit had deleted — VERDICT r4)."""
import csv
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_newest_kernel_statistics_name_only_kernels_that_exist():
    files = glob.glob(os.path.join(ROOT, "profiles", "r*_kernel_stats.csv"))
    assert files
    rnd = max(int(re.match(r"r(\d+)", os.path.basename(f)).group(1)) for f in files)
    newest = [f for f in files if re.match(r"r%d[a-z]?_" % rnd, os.path.basename(f))]
    src = ""
    csrc = os.path.join(ROOT, "webauthn-halo2_amd", "csrc")
    for f in os.listdir(csrc):
        src += open(os.path.join(csrc, f)).read()
    seen = set()
    for f in newest:
        for row in csv.DictReader(open(f)):
            m = re.search(r"zk::([A-Za-z0-9_]+)", row["Name"].split("(")[0])
            if m:
                seen.add(m.group(1))
    assert "msm_wacc_fast_kernel" in seen and len(seen) > 20
    missing = sorted(k for k in seen if not re.search(r"\b%s\b" % re.escape(k), src))
    assert not missing, "profiles/ r%d names kernels that csrc/ no longer has: %s" % (rnd, missing)
