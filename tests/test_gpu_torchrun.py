"""The N > 1 launch path of bench.py on real hardware at world size 1: under torchrun RANK is set, so the worker takes the
`nccl` (= RCCL) init, the CUDA barrier and the all_reduce(MAX) / all_gather of the clock exactly as an 8-GPU run does
(bench.py main()).  One worker per request/GPU is the reference's concurrency model (proving-server/src/main.rs:457-472);
proofs are replicas, there is no data-path collective to test."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bench_under_torchrun_one_rank_nccl():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "ZKMI355_BENCH_FAKE")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-1000:], out.stderr[-3000:])
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["metric"] == "webauthn_es256_proofs_per_sec_k19" and j["n_gpus"] == 1 and j["steps"] == 2
    assert j["launcher"] == "torchrun" and j["dist_backend"] == "nccl"
    assert len(j["ms_per_step_per_rank"]) == 1 and abs(j["ms_per_step_per_rank"][0] - j["ms_per_step"]) < 1e-6
    assert j["config"]["proof_bytes"] == 960 and j["value"] > 1.0
