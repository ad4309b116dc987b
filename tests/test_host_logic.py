"""Host-side logic that needs no GPU: ES256 request validation, the JSON request contract of the
reference's proving server (proving-server/src/main.rs:39-79), job assignment of the batch driver."""
import json
import random

import pytest

from webauthn_halo2_amd import batch, circuit, ecdsa_p256 as api, proving_server as srv


def sign(d, k, z):
    r = api._p256_mul(k, api._G)[0] % api._N
    return r, pow(k, -1, api._N) * (z + r * d) % api._N


LE = lambda v: v.to_bytes(32, "little")


def test_es256_verify_known_answer():
    # RFC 6979 A.2.5 (P-256, SHA-256, message "sample"): private key, public key and signature
    d = 0xC9AFA9D845BA75166B5C215767B1D6934E50C3DB36E89B127B8A622B120F6721
    qx = 0x60FED4BA255A9D31C961EB74C6356D68C049B8923B61FA6CE669622E60F29FB6
    qy = 0x7903FE1008B8BC99A41AE9E95628BC64F2F1B20C2D7E9F5177A3C294D4462299
    z = 0xAF2BDBE1AA9B6EC1E2ADE1D694F41FC71A831D0268E9891562113D8A62ADD1BF  # SHA-256("sample")
    r = 0xEFD48B2AACB6A8FD1140DD9CD45E81D69D2C877B56AAF991C34D0EA84EAF3716
    s = 0xF7CB1C942D657C41D436C7A1B6E29F65F3E900DBB9AFF4064DC4AB2F843ACDA8
    assert api._p256_mul(d, api._G) == (qx, qy)
    assert api.es256_verify(LE(qx), LE(qy), LE(r), LE(s), LE(z))
    assert not api.es256_verify(LE(qx), LE(qy), LE(r), LE(s ^ 1), LE(z))
    assert not api.es256_verify(LE(qx), LE(qy), LE(r), LE(s), LE(z ^ 1))
    assert not api.es256_verify(LE(qx), LE(qy ^ 1), LE(r), LE(s), LE(z))        # off the curve
    assert not api.es256_verify(LE(qx), LE(qy), LE(0), LE(s), LE(z))
    assert not api.es256_verify(LE(qx), LE(qy), LE(r), LE(api._N), LE(z))       # non-canonical scalar
    assert not api.es256_verify(LE(api._P), LE(qy), LE(r), LE(s), LE(z))        # non-canonical coordinate
    assert not api.es256_verify(*[bytes([i]) * 32 for i in range(5)])


def test_es256_random_signatures():
    rng = random.Random(7)
    for _ in range(5):
        d, k, z = (rng.randrange(1, api._N) for _ in range(3))
        q = api._p256_mul(d, api._G)
        r, s = sign(d, k, z)
        assert api.es256_verify(LE(q[0]), LE(q[1]), LE(r), LE(s), LE(z))
        assert not api.es256_verify(LE(q[0]), LE(q[1]), LE(r), LE(s), LE((z + 1) % api._N))


def test_request_shaped_entry_points_are_named_synthetic():
    assert not hasattr(api, "generate_proof") and not hasattr(api, "generate_proof_evm")
    assert "SYNTHETIC" in api.generate_proof_synthetic.__doc__ and "SYNTHETIC" in api.generate_proof_evm_synthetic.__doc__


def test_parse_request_is_as_strict_as_serde():
    body = {f: [i] * 32 for i, f in enumerate(srv.FIELDS)}
    body["proving_key_path"] = "./keys/proving_key.pk"
    q = srv.parse_request(json.dumps(body))
    assert q["msghash"] == bytes([4]) * 32 and q["proving_key_path"] == "./keys/proving_key.pk"
    assert srv.parse_request(body) == q
    for bad in (dict(body, r=[0] * 31), dict(body, r=[0] * 33), dict(body, s=[-1] + [0] * 31), dict(body, s=[256] + [0] * 31),
                dict(body, pubkey_x="00" * 32), dict(body, pubkey_y=[0.5] * 32), dict(body, msghash=[True] * 32),
                {k: v for k, v in body.items() if k != "proving_key_path"}, [body]):
        with pytest.raises(ValueError):
            srv.parse_request(bad)
    assert srv.DEGREE == 17


def test_job_assignment_and_seeds():
    jobs = list(range(256))
    for world in (1, 2, 4, 8):
        parts = [batch.assign(jobs, r, world) for r in range(world)]
        assert sorted(sum(parts, [])) == jobs and all(p == jobs[r::world] for r, p in enumerate(parts))
    assert batch.job_seed(5) == 0x5EED0019 + 5
    assert len({batch.job_rng_seed(i) for i in range(256)}) == 256 and all(len(batch.job_rng_seed(i)) == 32 for i in range(4))


def test_synthesize_jobs_matches_lone_synthesis():
    p = circuit.CircuitParams(degree=7, num_advice=2, num_lookup_advice=1, num_fixed=1, lookup_bits=5)
    got = batch.synthesize_jobs(p, [0, 3, 4], processes=2)
    for j in (0, 3, 4):
        asg = circuit.synthesize(p, batch.job_seed(j))
        assert len(got[j]) == len(asg.advice)
        assert all((a == asg.to_limbs(c)).all() for a, c in zip(got[j], asg.advice))
    fixed, copies = batch.structure(p)
    assert fixed.shape == (circuit.Layout(p).n_fix, 128, 4) and copies == circuit.synthesize(p, 99).copies


def test_stream_audit_ledger_logic(tmp_path):
    """csrc/audit.h (ZK_OPT_STREAM_AUDIT) on scripted enqueue sequences, host only: read-after-write / write-after-read across
    streams with and without the event pair, transitivity through a third stream, an event recorded before the producer, round 5's
    shared-scratch bug, host reads before and after the wait."""
    import os
    import shutil
    import subprocess

    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("hipcc") is None:
        import pytest

        pytest.skip("hipcc not on PATH")
    exe = str(tmp_path / "audit_logic_check")
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O1", "-std=c++17", "-x", "hip", "-I", os.path.join(ROOT, "webauthn-halo2_amd", "csrc"),
                           os.path.join(ROOT, "tests", "audit_logic_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "audit logic: 0 failures" in out.stdout, out.stdout + out.stderr
