#!/usr/bin/env python3
"""Proof latency of every row of the reference's bench_ecdsa.config on one GPU.

Mirrors the reference's `bench_secp256k1_ecdsa` loop (halo2-circuits/src/ecc/ecdsa_p256.rs:232-330:
per row: keygen, one Blake2b/SHPLONK proof timed, proof size, verify) with the synthetic same-shape
witness; prints one CSV line per row next to the published proof time of
halo2-circuits/src/results/ecdsa_bench.csv (another machine — context, not a baseline).
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import webauthn_halo2_amd as zk  # noqa: E402
from webauthn_halo2_amd import engine as E  # noqa: E402

# degree, num_advice, num_lookup_advice, num_fixed, lookup_bits, idle gate columns, published seconds, published bytes
ROWS = [
    (19, 1, 1, 1, 18, 0, 14.846, 960),
    (18, 2, 1, 1, 17, 0, 8.908, 1344),
    (17, 4, 1, 1, 16, 0, 5.388, 1920),
    (16, 8, 2, 1, 15, 0, 4.377, 3552),
    (15, 17, 3, 1, 14, 0, 4.134, 6560),
    (14, 34, 6, 1, 13, 0, 4.170, 12704),
    (13, 68, 12, 1, 12, 1, 4.671, 24960),
    (12, 139, 24, 2, 11, 2, 5.507, 50496),
    (11, 291, 53, 4, 10, 3, 6.605, 106496),
]


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    eng = zk.Engine(0)
    if os.environ.get("MSM_WINDOW"):  # tuning runs: force the fixed-base MSM window for every row
        eng.set_option(E.ZK_OPT_MSM_WINDOW, int(os.environ["MSM_WINDOW"]))
    for o in os.environ.get("OPTS", "").split(","):  # OPTS=13=1: zk_ctx_set_option
        if o:
            eng.set_option(*(int(x) for x in o.split("=")))
    only = [int(x) for x in os.environ["ROWS"].split(",")] if os.environ.get("ROWS") else None
    print("degree,num_advice,num_lookup,num_fixed,lookup_bits,proof_ms,proof_size,published_cpu_s,speedup")
    for k, A, L, F, lb, idle, pub_s, pub_b in ROWS:
        if only and k not in only:
            continue
        p = zk.circuit.CircuitParams(degree=k, num_advice=A, num_lookup_advice=L, num_fixed=F, lookup_bits=lb,
                                     idle_gate_columns=idle)
        asg = zk.circuit.synthesize(p, 0x5EED0019)
        eng.srs_setup(k)
        pk = eng.keygen(p, np.stack([asg.to_limbs(c) for c in asg.fixed]), asg.copies)
        polys = []
        for col in asg.advice:
            h = eng.poly(1 << k)
            eng.upload_canonical(h, asg.to_limbs(col))
            polys.append(h)
        eng.prove(pk, polys, bytes(32), E.ZK_TRANSCRIPT_BLAKE2B)  # warm-up
        best = 1e9
        for r in range(reps):
            t0 = time.perf_counter()
            pf = eng.prove(pk, polys, bytes([r + 1]) * 32, E.ZK_TRANSCRIPT_BLAKE2B)
            best = min(best, time.perf_counter() - t0)
        assert len(pf) == pub_b, (k, len(pf), pub_b)
        print(f"{k},{A},{L},{F},{lb},{best * 1e3:.2f},{len(pf)},{pub_s},{pub_s / best:.0f}", flush=True)
        for h in polys:
            h.free()
        eng.pk_free(pk)


if __name__ == "__main__":
    main()
