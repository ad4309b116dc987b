#!/usr/bin/env python
"""bench.py — headline benchmark: WebAuthn ES256 (secp256r1 ECDSA circuit) proofs/sec at k=19.

One "step" = one pass of the create_proof hot path over one synthetic witness of
the k=19 shape (BASELINE.json configs[1]), inputs resident in HBM.
N>1: independent proofs, one per GPU (replicas only, no collective on the data
path — SURVEY.md §8e); torch.distributed (RCCL) is used for the barrier and the
max-over-ranks clock only.

Prints ONE JSON line on rank 0 (contract in the task statement), including
`roofline` for the dominant kernel (HIP events on the engine's own stream) and
`cpu_baseline` (the oracle's C restatement of best_multiexp / best_fft timed on
the host cores; the reference Rust prover cannot be built here — no toolchain).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "oracle")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402

K = 19
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def witness_like_scalars(n, seed):
    """The advice-column mix of SURVEY.md §8d (canonical ints -> bytes, *not* Montgomery:
    only the distribution matters for operator timing)."""
    rng = np.random.default_rng(seed)
    a = np.zeros((n, 4), dtype=np.uint64)
    u = rng.random(n)
    small = u < 0.40
    mid = (u >= 0.40) & (u < 0.75)
    full = (u >= 0.75) & (u < 0.90)
    a[small, 0] = rng.integers(0, 1 << 18, small.sum(), dtype=np.uint64)
    a[mid, 0] = rng.integers(0, 1 << 63, mid.sum(), dtype=np.uint64)
    a[mid, 1] = rng.integers(0, 1 << 24, mid.sum(), dtype=np.uint64)
    r = rng.integers(0, 1 << 62, (int(full.sum()), 4), dtype=np.uint64)
    r[:, 3] &= np.uint64(0x0FFFFFFFFFFFFFFF)
    a[full] = r
    return a


class OperatorWorkload:
    """k=19 Blake2b/SHPLONK-shape operator sequence of one proof:
    12 MSM(2^19) + 5 iNTT(2^19) + 5 coset-NTT(2^21) + 1 inverse coset-NTT(2^21) + 18 evaluations.
    Used until the full device prover lands; named as such in config.workload."""

    name = "k19-operator-sequence:12xMSM(2^19)+5xiNTT(2^19)+5xcosetNTT(2^21)+1xcosetiNTT(2^21)+18xeval(2^19)"

    def __init__(self, eng, seed):
        self.eng = eng
        n = 1 << K
        eng.srs_setup(K)
        uni = np.frombuffer(np.random.default_rng(seed).bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
        uni[:, 3] &= 0x0FFFFFFFFFFFFFFF
        self.advice = eng.poly(n, witness_like_scalars(n, seed))
        self.uniform = eng.poly(n, uni)
        self.work = eng.poly(n)
        self.ext = [eng.poly(4 * n) for _ in range(2)]
        self.x = uni[7].copy()
        self.accum_ms = []
        self.msm_ms = []

    def _commit(self, p, basis):
        e = self.eng
        e.commit(p, basis)
        self.accum_ms.append(e.last_ms(4))
        self.msm_ms.append(e.last_ms(0))

    def step(self):
        e = self.eng
        # phase 1-4: advice, a', s', z, zL  -> commit_lagrange (5)
        self._commit(self.advice, 1)
        for _ in range(4):
            self._commit(self.uniform, 1)
        # phase 5: random poly commit (monomial)
        self._commit(self.uniform, 0)
        # iNTTs of advice, a', s', z, zL; coset NTTs of the same five
        for i in range(5):
            e.copy(self.work, self.uniform if i else self.advice)
            e.lagrange_to_coeff(self.work)
            e.coeff_to_extended(self.work, self.ext[i & 1])
        # quotient: inverse coset NTT, 4 h pieces
        e.extended_to_coeff(self.ext[0], 4 << K)
        for _ in range(4):
            self._commit(self.uniform, 0)
        for _ in range(18):
            e.eval(self.uniform, self.x)
        # multi-open: 2 commits (SHPLONK)
        for _ in range(2):
            self._commit(self.uniform, 0)
        e.sync()


def cpu_baseline(eng):
    """Oracle C restatement (halo2 best_multiexp / best_fft) on the host cores:
    one MSM(2^19) + one NTT(2^19) + one NTT(2^21), scaled to the per-proof operator counts."""
    from zkoracle import cops, field as F

    n = 1 << K
    cores = os.cpu_count() or 1
    bases = eng.srs_export(0, 0, n)
    s = np.frombuffer(np.random.default_rng(5).bytes(n * 32), dtype=np.uint64).reshape(n, 4).copy()
    s[:, 3] &= 0x0FFFFFFFFFFFFFFF
    t0 = time.time()
    cops.msm(s, bases, cores)
    t_msm = time.time() - t0
    t0 = time.time()
    cops.ntt(s, F.omega(K), K, cores)
    t_ntt19 = time.time() - t0
    big = np.concatenate([s, s, s, s])
    t0 = time.time()
    cops.ntt(big, F.omega(K + 2), K + 2, cores)
    t_ntt21 = time.time() - t0
    per_proof = 12 * t_msm + 5 * t_ntt19 + 6 * t_ntt21
    return {
        "value": 1.0 / per_proof,
        "unit": "proofs/s",
        "cores": cores,
        "kind": "port",
        "sample": "oracle C port of halo2 best_multiexp/best_fft: 1xMSM(2^19)=%.2fs, 1xNTT(2^19)=%.3fs, 1xNTT(2^21)=%.3fs; "
        "scaled to 12 MSM + 5 NTT(2^19) + 6 NTT(2^21) per proof (quotient/eval not included); "
        "the reference Rust prover cannot be built on this node (no cargo/rustc)" % (t_msm, t_ntt19, t_ntt21),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    torch = None
    if world > 1:
        import torch
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    import webauthn_halo2_amd as zk

    eng = zk.Engine(local_rank)
    wl = OperatorWorkload(eng, 0x5EED0019 + rank)

    def barrier():
        eng.sync()
        if dist is not None:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        wl.step()
    wl.accum_ms.clear()
    wl.msm_ms.clear()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        n = 1 << K
        accum_ms = float(np.mean(wl.accum_ms))
        alg_bytes = 96.0 * n  # MSM(n) = 32 B scalar + 64 B base per point
        achieved = alg_bytes / (accum_ms * 1e-3) / 1e9
        out = {
            "metric": "webauthn_es256_proofs_per_sec_k19",
            "value": world * args.steps / elapsed,
            "unit": "proofs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u256-montgomery(8x32-bit limbs)",
            "data": "synthetic",
            "config": {"workload": wl.name, "k": K, "transcript": "none (operator sequence)", "parallelism": "replicas:%d" % world},
            "roofline": {
                "kernel": "msm_accumulate_kernel",
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": None,
                "avg_launch_ms": accum_ms,
                "note": "integer-ALU-bound kernel (no MFMA, SURVEY.md §8d); whole-MSM avg %.3f ms" % float(np.mean(wl.msm_ms)),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(eng)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
