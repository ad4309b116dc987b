"""Which hardware queue each of the engine's streams lands on (rocprofv3 --kernel-trace, Queue_Id): four contexts are made one after
the other; then every context commits one column ALONE (main stream: sort head + accumulation; side stream: the tail), phases
50 ms apart; then context 0 proves a lone k = 18 proof (main, tail, transform and MSM streams).
    run:    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python tools/queue_map.py run
    report: python tools/queue_map.py report <kernel_trace.csv>"""
import csv
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import numpy as np
    import webauthn_halo2_amd as zk
    from webauthn_halo2_amd import engine as E
    k = 18
    n = 1 << k
    engs = [zk.Engine(0)]
    engs[0].srs_setup(k)
    for _ in range(3):
        engs.append(zk.Engine(0, share_with=engs[0]))
    a = np.frombuffer(np.random.default_rng(1).bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
    a[:, 3] &= 0x0FFFFFFFFFFFFFFF
    polys = [e.poly(n, a) for e in engs]
    time.sleep(0.2)
    for e, p in zip(engs, polys):
        e.commit(p, E.ZK_BASIS_LAGRANGE)
        e.sync()
        time.sleep(0.05)
    p = zk.circuit.CircuitParams(degree=k, num_advice=2, num_lookup_advice=1, num_fixed=1, lookup_bits=17)
    asg = zk.circuit.synthesize(p, 1)
    pk = engs[0].keygen(p, np.stack([asg.to_limbs(c) for c in asg.fixed]), asg.copies)
    cols = []
    for col in asg.advice:
        h = engs[0].poly(n)
        engs[0].upload_canonical(h, asg.to_limbs(col))
        cols.append(h)
    time.sleep(0.2)
    engs[0].set_option(E.ZK_OPT_XFORM_STREAM, 1)  # (k = 18 with two columns: force the lone regime's streams)
    engs[0].set_option(E.ZK_OPT_MSM_STREAM, 1)
    engs[0].prove(pk, cols, bytes(32), E.ZK_TRANSCRIPT_BLAKE2B)
    engs[0].sync()


def report(path):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    phases, last = [[]], None
    for r in rows:
        t = int(r["Start_Timestamp"])
        if last is not None and t - last > 30_000_000:
            phases.append([])
        phases[-1].append(r)
        last = int(r["End_Timestamp"])
    for i, ph in enumerate(phases):
        q = {}
        for r in ph:
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("zk::", "")
            q.setdefault(r["Queue_Id"], {}).setdefault(name.split("<")[0], 0)
            q[r["Queue_Id"]][name.split("<")[0]] += 1
        print("phase %d: %d kernels" % (i, len(ph)))
        for qi in sorted(q):
            top = sorted(q[qi].items(), key=lambda kv: -kv[1])[:5]
            print("   queue %s: %s" % (qi, ", ".join("%s x%d" % kv for kv in top)))


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else report(sys.argv[2])
