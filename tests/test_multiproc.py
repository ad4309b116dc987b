"""N>1 path on CPU: bench.py launched exactly as the driver does (torch.distributed.run, one
process per rank, 127.0.0.1 rendezvous) with the gloo backend and the fake workload — covers the
rank/env handling, the barrier-bracketed timing and the max-over-ranks reduction.  Proofs are
independent per rank (replicas, no data-path collective), so there is nothing else to exchange."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bench_two_ranks_gloo():
    env = dict(os.environ, ZKMI355_BENCH_FAKE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1  # only rank 0 prints
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 5 and j["scaling"] == "weak"
    # rank 1 sleeps 20 ms per step: the max over ranks must dominate (>= 5 * 20 ms)
    assert j["ms_per_step"] >= 19.0
    assert abs(j["value"] - 2 * 5 / (j["ms_per_step"] * 5 / 1e3)) < 1e-6


def test_bench_launches_its_own_workers():
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE: the script itself starts one worker per
    device (torch.distributed.run on 127.0.0.1) and relays rank 0's single JSON line — what the driver's 8-GPU run
    relies on.  Fake workload (no GPU here): rank r sleeps 10 (r + 1) ms per job."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "ZKMI355_BENCH_WORKER")}
    env["ZKMI355_BENCH_FAKE"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1"], env=env,
                         capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 4 and j["jobs_total"] == 8
    assert j["ms_per_step"] >= 19.0  # the slower rank sets the clock
    # --gpus 1 stays in-process
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1"], env=env,
                         capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0 and json.loads(out.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_job_seeds_are_disjoint_across_ranks():
    # BASELINE config 4: job i uses witness seed 0x5eed0019 + i; rank r of N takes i = r, r + N, ...
    import webauthn_halo2_amd as zk
    from webauthn_halo2_amd import batch

    jobs = list(range(256))
    for n in (1, 2, 4, 8):
        seen = []
        for r in range(n):
            mine = batch.assign(jobs, r, n)
            assert mine == jobs[r::n]
            seen += mine
        assert sorted(seen) == jobs
    assert batch.job_seed(3) == 0x5EED0019 + 3
    del zk


def test_eight_ranks_cover_the_256_job_batch_exactly_once():
    """BASELINE configs[3] as the driver launches it on an 8-GPU node (`--gpus 8 --steps 32`), on CPU with the fake workload
    (gloo, world size 8): the ranks' jobs partition 0 .. 255 — every job exactly once —, every one of them has a committed oracle
    digest for its rank to compare the proof with (tests/golden/batch_k19_sha256.json), every rank's own clock and set-up time
    travel in the line, and the slowest rank sets `value`."""
    env = dict(os.environ, ZKMI355_BENCH_FAKE="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "32", "--warmup", "1"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["steps"] == 32 and j["jobs_total"] == 256
    assert j["jobs_covered_exactly_once"] is True
    assert j["proofs_checked_against_oracle_digests"] == 256
    assert len(j["ms_per_step_per_rank"]) == 8 and len(j["host_setup"]["keygen_s_per_rank"]) == 8
    assert j["ms_per_step"] >= max(j["ms_per_step_per_rank"]) - 1e-6  # rank 7 sleeps 80 ms per job
