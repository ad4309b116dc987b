"""examples/prove_host.cpp — a plain C++ host of the C ABI (the reference's per-request path: read params, read the proving
key, create_proof; halo2-circuits/src/ecc/ecdsa_p256.rs:388-428) — produces the proof bytes the ctypes binding produces for
the same key files, witness and RNG seed, and the oracle accepts them."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import webauthn_halo2_amd as zk  # noqa: E402
from webauthn_halo2_amd import engine as E  # noqa: E402
from zkoracle import cops, plonk  # noqa: E402

HOST = os.path.join(ROOT, "examples", "prove_host")


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["blake2b", "evm"])
def test_cpp_host_proves_what_the_binding_proves(tmp_path, kind):
    if not os.path.exists(HOST):  # normally built by build.sh / __graft_entry__.build(); plain g++, seconds
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "prove_host.cpp"),
                               "-L" + os.path.join(ROOT, "webauthn-halo2_amd"), "-lzkmi355",
                               "-Wl,-rpath," + os.path.join(ROOT, "webauthn-halo2_amd"), "-o", HOST])
    k, A, L, F, lb = 10, 3, 2, 1, 8
    p = zk.circuit.CircuitParams(degree=k, num_advice=A, num_lookup_advice=L, num_fixed=F, lookup_bits=lb)
    asg = zk.circuit.synthesize(p, 0x5EED0019)
    eng = zk.Engine(0)
    eng.srs_setup(k)
    pk = eng.keygen(p, np.stack([asg.to_limbs(c) for c in asg.fixed]), asg.copies)
    (tmp_path / "srs.bin").write_bytes(eng.srs_write(E.ZK_SERDE_RAW_BYTES))
    (tmp_path / "pk.bin").write_bytes(eng.pk_write(pk, E.ZK_SERDE_RAW_BYTES))
    cols = np.stack([asg.to_limbs(c) for c in asg.advice])
    (tmp_path / "advice.bin").write_bytes(np.ascontiguousarray(cols, dtype="<u8").tobytes())
    seed = bytes(range(7, 39))
    polys = []
    for col in cols:
        h = eng.poly(1 << k)
        eng.upload_canonical(h, col)
        polys.append(h)
    tk = E.ZK_TRANSCRIPT_EVM if kind == "evm" else E.ZK_TRANSCRIPT_BLAKE2B
    want = eng.prove(pk, polys, seed, tk)
    fc, pc, tr = eng.vk_export(pk)
    eng.close()  # the host below is a process of its own with its own context
    out = tmp_path / "proof.bin"
    r = subprocess.run([HOST, str(tmp_path / "srs.bin"), str(tmp_path / "pk.bin"), str(tmp_path / "advice.bin"), str(out),
                        str(k), str(A), str(L), str(F), str(lb), "0", kind, seed.hex()], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    got = out.read_bytes()
    assert got == want
    vk = plonk.VerifyingKey(plonk.Shape(k, A, L, F, lb, 0), cops.affine_arr_to_ints(fc), cops.affine_arr_to_ints(pc),
                            cops.fr_ints(tr.reshape(1, 4))[0])
    assert plonk.verify(vk, got, kind)
    # a truncated key file is refused, nothing is written
    (tmp_path / "pk_short.bin").write_bytes((tmp_path / "pk.bin").read_bytes()[:-64])
    bad = subprocess.run([HOST, str(tmp_path / "srs.bin"), str(tmp_path / "pk_short.bin"), str(tmp_path / "advice.bin"),
                          str(tmp_path / "none.bin"), str(k), str(A), str(L), str(F), str(lb), "0", kind, seed.hex()],
                         capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0 and "zk_pk_read" in bad.stderr and not (tmp_path / "none.bin").exists()
