#!/usr/bin/env python3
"""Bug hunt (test infrastructure: it uses the oracle, so it lives under tests/): proofs of randomly drawn circuit shapes,
device against the oracle.
  shape_sweep.py <seed> <count>          small shapes (k 6..9) against the plain-Python oracle — the committed test
                                         tests/test_gpu_prover.py::test_random_shapes_byte_identical_to_oracle runs ten of them
  shape_sweep.py <seed> <count> mid      k 10..14 with up to ~90 columns against the oracle's numpy + C prover (fastprover)
  shape_sweep.py <seed> <count> full     the bench_ecdsa.config rows k = 16..19 at full size, `count` witness seeds each"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))  # this directory: the suite's helpers
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import test_gpu_prover as t  # noqa: E402
from zkoracle import fastprover as fp  # noqa: E402


def mid_shapes(count, seed):
    pr = random.Random(seed)
    out = []
    while len(out) < count:
        k = pr.choice([10, 11, 12, 13, 14])
        A = pr.choice([1, 2, 3, 4, 8, 17, 34, 68])
        if A * (1 << k) > 68 << 12:      # bound the oracle's time
            continue
        L = 1 if A == 1 else pr.choice([1, 2, max(1, A // 6), max(1, A // 4)])
        F = pr.choice([1, 1, 2, 4])
        lb = pr.randrange(k - 4, k)
        idle = min(pr.choice([0, 0, 0, 1, 3]) if A >= 4 else 0, A // 2)
        out.append((A, L, F, k, lb, idle))
    return out


def check_mid(eng, shape, wseed=None):
    A, L, F, k, lb, idle = shape
    p, asg, pk, polys = t.setup(eng, A, L, F, k, lb, seed=wseed if wseed is not None else 0x5EED1000 + 131 * A + k, idle=idle)
    sh = t.plonk.Shape(k, A, L, F, lb, idle)
    fpk = fp.keygen(sh, asg.fixed, asg.copies)
    vk = t.product_vk(eng, pk, sh)
    assert vk.fixed_commitments == fpk.vk.fixed_commitments and vk.permutation_commitments == fpk.vk.permutation_commitments
    assert vk.transcript_repr == fpk.vk.transcript_repr
    seed = bytes([k, A & 255, L, F]) * 8
    for kind in ("blake2b", "evm"):
        got = eng.prove(pk, polys, seed, t.KIND[kind])
        want = fp.create_proof(fpk, asg.advice, t.ChaCha20Rng(seed), kind)
        assert got == want, (shape, kind)
    for h in polys:
        h.free()
    eng.pk_free(pk)


def check_batch(eng, shape, wseed=None, B=3):
    """zk_prove_batch (lock-step) against zk_prove — whose bytes the checks above pin to the oracle — on B witnesses of the shape."""
    import numpy as np

    A, L, F, k, lb, idle = shape
    n_adv = A if A == 1 else A + L
    nprod = -(-(F + n_adv) // (3 if A == 1 else 2)) + (1 if A == 1 else L)
    B = min(B, 256 // nprod)  # all grand products of a batch go through one 256-lane chain scan
    if B < 2:
        return
    p = t.zk.circuit.CircuitParams(degree=k, num_advice=A, num_lookup_advice=L, num_fixed=F, lookup_bits=lb, idle_gate_columns=idle)
    base = wseed if wseed is not None else 0x5EED2000 + 977 * A + k
    asgs = [t.zk.circuit.synthesize(p, base + 13 * i) for i in range(B)]
    eng.srs_setup(k)
    pk = eng.keygen(p, np.stack([asgs[0].to_limbs(c) for c in asgs[0].fixed]), asgs[0].copies)
    sets = []
    for asg in asgs:
        hs = []
        for col in asg.advice:
            h = eng.poly(1 << k)
            eng.upload_canonical(h, asg.to_limbs(col))
            hs.append(h)
        sets.append(hs)
    seeds = [bytes([k, (A + i) & 255, L, F]) * 8 for i in range(B)]
    try:
        for kind in ("blake2b", "evm"):
            got = eng.prove_batch(pk, sets, seeds, t.KIND[kind])
            for j in range(B):
                assert got[j] == eng.prove(pk, sets[j], seeds[j], t.KIND[kind]), (shape, kind, j)
    finally:
        for hs in sets:
            for h in hs:
                h.free()
        eng.pk_free(pk)


def main():
    seed, count = int(sys.argv[1], 0), int(sys.argv[2])
    mid = len(sys.argv) > 3 and sys.argv[3] == "mid"
    eng = t.zk.Engine(0)
    for o in os.environ.get("OPTS", "").split(","):  # OPTS=13=2: zk_ctx_set_option (e.g. the three-coset quotient on every shape it applies to)
        if o:
            eng.set_option(*(int(x) for x in o.split("=")))
    bad = 0
    if len(sys.argv) > 3 and sys.argv[3] == "full":
        for shape in ((1, 1, 1, 19, 18, 0), (2, 1, 1, 18, 17, 0), (4, 1, 1, 17, 16, 0), (8, 2, 1, 16, 15, 0)):
            for i in range(count):
                t0 = time.time()
                try:
                    check_mid(eng, shape, wseed=seed + 1000 * shape[3] + i)
                    check_batch(eng, shape, wseed=seed + 1000 * shape[3] + i)
                    print("ok  ", shape, "witness", i, "%.1f s" % (time.time() - t0), flush=True)
                except Exception as e:  # noqa: BLE001
                    bad += 1
                    print("FAIL", shape, i, repr(e)[:300], flush=True)
        print("failures", bad)
        return 1 if bad else 0
    shapes = mid_shapes(count, seed) if mid else t._random_shapes(count, seed)
    for shape in shapes:
        t0 = time.time()
        try:
            (check_mid if mid else t.test_random_shapes_byte_identical_to_oracle)(eng, shape)
            check_batch(eng, shape if mid else (shape + (0,))[:6])
            print("ok  ", shape, "%.1f s" % (time.time() - t0), flush=True)
        except Exception as e:  # noqa: BLE001
            bad += 1
            print("FAIL", shape, repr(e)[:300], flush=True)
    print("shapes", count, "failures", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
