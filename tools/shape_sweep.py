#!/usr/bin/env python3
"""Bug hunt: proofs of randomly drawn circuit shapes, device against the plain-Python oracle (the committed test
tests/test_gpu_prover.py::test_random_shapes_byte_identical_to_oracle runs ten of them; this draws as many as asked).
usage: shape_sweep.py <seed> <count>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import test_gpu_prover as t  # noqa: E402


def main():
    seed, count = int(sys.argv[1], 0), int(sys.argv[2])
    eng = t.zk.Engine(0)
    bad = 0
    for shape in t._random_shapes(count, seed):
        try:
            t.test_random_shapes_byte_identical_to_oracle(eng, shape)
            print("ok  ", shape, flush=True)
        except Exception as e:  # noqa: BLE001
            bad += 1
            print("FAIL", shape, repr(e)[:200], flush=True)
    print("shapes", count, "failures", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
