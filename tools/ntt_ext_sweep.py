"""coeff_to_extended (2^k -> 2^(k+2), the zero-padded ZQ first pass) and extended_to_coeff by largest radix (zk_ctx_set_option 3):
which plan the proof's big transforms want.  usage: ntt_ext_sweep.py [k ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import webauthn_halo2_amd as zk
from webauthn_halo2_amd import engine as E

eng = zk.Engine(0)
for k in [int(x) for x in (sys.argv[1:] or ["19", "17"])]:
    n, N = 1 << k, 4 << k
    col = np.random.default_rng(1).integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    col[:, 3] &= np.uint64((1 << 60) - 1)
    p, ext = eng.poly(n, col), eng.poly(N)
    fw, bw = [], []
    for r in (0, 6, 7, 8, 9, 10, 11):
        eng.set_option(E.ZK_OPT_NTT_MAX_RADIX_LOG2, r)
        t = []
        for _ in range(5):
            eng.coeff_to_extended(p, ext); eng.sync(); t.append(eng.last_ms(E.ZK_T_NTT))
        fw.append("r%d:%.3f" % (r, min(t)))
        t = []
        for _ in range(5):
            eng.extended_to_coeff(ext, N); eng.sync(); t.append(eng.last_ms(E.ZK_T_NTT))
        bw.append("r%d:%.3f" % (r, min(t)))
    print("2^%d -> 2^%d  to extended  %s" % (k, k + 2, "  ".join(fw)))
    print("2^%d -> 2^%d  back         %s" % (k, k + 2, "  ".join(bw)))
    p.free(); ext.free()
