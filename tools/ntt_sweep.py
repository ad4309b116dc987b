"""NTT time at 2^21 / 2^19 as a function of the largest pass radix (zk_ctx_set_option) — run under each tile-size
build (ZKMI355_LIB, tools/ab_variants.sh) to see which (radix, columns-per-tile) shapes the memory system likes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import webauthn_halo2_amd as zk
from webauthn_halo2_amd import engine as E

eng = zk.Engine(0)
for k in [int(x) for x in os.environ.get("KS", "21,19").split(",")]:
    n = 1 << k
    a = np.frombuffer(np.random.default_rng(1).bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
    a[:, 3] &= 0x0FFFFFFFFFFFFFFF
    p = eng.poly(n, a)
    out = []
    for r in range(4, 12):
        try:
            eng.set_option(E.ZK_OPT_NTT_MAX_RADIX_LOG2, r)
        except zk.ZkError:
            continue
        ts = []
        for _ in range(6):
            eng.coeff_to_lagrange(p); eng.sync(); ts.append(eng.last_ms(1))
        out.append("r%d:%.3f" % (r, min(ts)))
    print("2^%d  %s" % (k, "  ".join(out)))
    p.free()
