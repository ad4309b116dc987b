"""examples/prove_host_phases.cpp — a host that owns the transcript, the RNG and the blinding and builds a whole proof from the
PHASE-LEVEL C ABI only (zk_commit_batch, zk_lookup_permute, zk_lookup_product, zk_permutation_product, zk_random_poly,
zk_lagrange_to_coeff, zk_coeff_to_extended, zk_quotient, zk_extended_to_coeff, zk_eval, zk_poly_lincomb, zk_kate_division), with
either multi-open scheme (ProverGWC, ProverSHPLONK): the resident integration of INTEGRATION.md §2 executed.  Its bytes are zk_prove's for the same key, advice and
seed — and, at the proving server's k = 17 shape with the EVM transcript + GWC, the fixture the reference's own Yul verifier
accepts (tests/golden/engine_proof_k17_evm.json, tests/test_oracle_verifier.py)."""
import json
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

HOST = os.path.join(ROOT, "examples", "prove_host_phases")
pytestmark = pytest.mark.gpu


def _run(tmp_path, p, seed, witness_seed, kinds):
    assert os.path.exists(HOST), "examples/prove_host_phases is built by build.sh / __graft_entry__.build()"
    asg = zk.circuit.synthesize(p, witness_seed)
    eng = zk.Engine(0)
    eng.srs_setup(p.degree)
    pk = eng.keygen(p, np.stack([asg.to_limbs(c) for c in asg.fixed]), asg.copies)
    (tmp_path / "srs.bin").write_bytes(eng.srs_write(E.ZK_SERDE_RAW_BYTES))
    (tmp_path / "pk.bin").write_bytes(eng.pk_write(pk, E.ZK_SERDE_RAW_BYTES))
    cols = np.stack([asg.to_limbs(c) for c in asg.advice])
    (tmp_path / "advice.bin").write_bytes(np.ascontiguousarray(cols, dtype="<u8").tobytes())
    polys = []
    for col in cols:
        h = eng.poly(1 << p.degree)
        eng.upload_canonical(h, col)
        polys.append(h)
    scheme_id = {"gwc": E.ZK_SCHEME_GWC, "shplonk": E.ZK_SCHEME_SHPLONK}
    want = {(kind, scheme): eng.prove(pk, polys, seed, E.ZK_TRANSCRIPT_EVM if kind == "evm" else E.ZK_TRANSCRIPT_BLAKE2B, scheme_id[scheme])
            for kind, scheme in kinds}
    eng.close()  # the host below is a process of its own with its own context
    got = {}
    for kind, scheme in kinds:
        out = tmp_path / ("proof_%s_%s.bin" % (kind, scheme))
        r = subprocess.run([HOST, str(tmp_path / "srs.bin"), str(tmp_path / "pk.bin"), str(tmp_path / "advice.bin"), str(out),
                            str(p.degree), str(p.num_advice), str(p.num_lookup_advice), str(p.num_fixed), str(p.lookup_bits),
                            str(p.idle_gate_columns), kind, seed.hex(), scheme], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        got[(kind, scheme)] = out.read_bytes()
    return got, want


@pytest.mark.parametrize("shape", [(10, 3, 2, 1, 8, 0), (10, 1, 1, 1, 9, 0), (8, 5, 2, 2, 6, 2)],
                         ids=["multi-column", "one-column", "combined-selectors"])
def test_phase_level_host_proves_what_zk_prove_proves(tmp_path, shape):
    k, A, L, F, lb, idle = shape
    p = zk.circuit.CircuitParams(degree=k, num_advice=A, num_lookup_advice=L, num_fixed=F, lookup_bits=lb, idle_gate_columns=idle)
    # the reference's pairings (EVM + GWC: /prove_evm; Blake2b + SHPLONK: /prove and the bench) and the two crossed ones
    got, want = _run(tmp_path, p, bytes(range(7, 39)), 0x5EED0019,
                     (("evm", "gwc"), ("blake2b", "shplonk"), ("blake2b", "gwc"), ("evm", "shplonk")))
    for key in got:
        assert got[key] == want[key], key


def test_phase_level_host_reproduces_the_yul_accepted_k17_proof(tmp_path):
    d = json.load(open(os.path.join(ROOT, "tests", "golden", "engine_proof_k17_evm.json")))
    got, want = _run(tmp_path, zk.circuit.K17, bytes.fromhex(d["rng_seed"]), 0x5EED0019, (("evm", "gwc"), ("blake2b", "shplonk")))
    assert len(got[("evm", "gwc")]) == 2720 and len(got[("blake2b", "shplonk")]) == 1920  # ecdsa_bench.csv:4: both sizes of the k = 17 row
    assert got == want
    assert got[("evm", "gwc")].hex() == d["proof"]
