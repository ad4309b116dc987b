"""csrc/hostutil.h's host-side inversion (binary extended Euclid, what the prover's host code calls between GPU phases) against the
Fermat inversion of field.hip.h on 40 000 elements of Fr and Fq, and the cached root-of-unity table against its definition — host
code only: hipcc compiles it here without a GPU (the header carries HIP's function attributes)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_fast_host_inverse_equals_fermat(tmp_path):
    exe = str(tmp_path / "host_inv_check")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-x", "hip", "-I", os.path.join(ROOT, "webauthn-halo2_amd", "csrc"),
                           os.path.join(ROOT, "tests", "host_inv_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "total bad 0" in out.stdout
