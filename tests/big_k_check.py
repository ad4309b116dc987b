#!/usr/bin/env python3
"""Beyond the BASELINE sizes (test infrastructure: uses the oracle's verifier): one proof of the k = 19 column shape at k = 20
and k = 21 (extended domains 2^22 / 2^23), both transcripts — proof sizes as the model gives them, accepted by the oracle's
verifier, deterministic.  usage: big_k_check.py [k ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import webauthn_halo2_amd as zk  # noqa: E402
from webauthn_halo2_amd import engine as E  # noqa: E402
from zkoracle import cops, plonk  # noqa: E402

for k in [int(x) for x in (sys.argv[1:] or ["20", "21"])]:
    A, L, F, lb = 1, 1, 1, k - 1
    p = zk.circuit.CircuitParams(degree=k, num_advice=A, num_lookup_advice=L, num_fixed=F, lookup_bits=lb)
    t0 = time.time()
    asg = zk.circuit.synthesize(p, 0x5EED0019)
    eng = zk.Engine(0)
    eng.srs_setup(k)
    pk = eng.keygen(p, np.stack([asg.to_limbs(c) for c in asg.fixed]), asg.copies)
    polys = []
    for col in asg.advice:
        h = eng.poly(1 << k)
        eng.upload_canonical(h, asg.to_limbs(col))
        polys.append(h)
    fc, pc, tr = eng.vk_export(pk)
    vk = plonk.VerifyingKey(plonk.Shape(k, A, L, F, lb, 0), cops.affine_arr_to_ints(fc), cops.affine_arr_to_ints(pc),
                            cops.fr_ints(tr.reshape(1, 4))[0])
    setup_s = time.time() - t0
    for kind, tk, size in (("blake2b", E.ZK_TRANSCRIPT_BLAKE2B, 960), ("evm", E.ZK_TRANSCRIPT_EVM, 1536)):
        eng.prove(pk, polys, b"\x05" * 32, tk)
        t1 = time.perf_counter()
        pf = eng.prove(pk, polys, b"\x05" * 32, tk)
        ms = (time.perf_counter() - t1) * 1e3
        assert len(pf) == size, (k, kind, len(pf))
        assert pf == eng.prove(pk, polys, b"\x05" * 32, tk)
        assert plonk.verify(vk, pf, kind), (k, kind)
        print(f"k={k} {kind}: {len(pf)} bytes in {ms:.1f} ms, accepted by the oracle verifier (set-up {setup_s:.1f} s)", flush=True)
    eng.close()
