"""Host-side mirror of the reference's proving API (halo2-circuits/src/ecc/ecdsa_p256.rs):
same function names, argument meaning and error behaviour, on top of the resident engine.

    download_keys(degree, proving_key_path, verifying_key_path)      ecdsa_p256.rs:256-272
    generate_proof(pubkey_x, pubkey_y, r, s, msg_hash, pk_path, k)   ecdsa_p256.rs:379-427  (Blake2b + SHPLONK)
    generate_proof_evm(...)                                          ecdsa_p256.rs:329-377  (EvmTranscript + GWC)

Differences that the scope of this repository imposes:
  * the reference re-reads the SRS and the proving key from disk on EVERY request
    (ecdsa_p256.rs:338-343); here `gen_srs` and the key stay resident on the device
     — the path arguments select a cached, resident key;
  * the secp256r1 witness generation (`ECDSACircuit::synthesize`, ecdsa_p256.rs:117-206) needs
    the Rust halo2-ecc chips and stays on the host side of the real integration; here the five
    32-byte little-endian request fields seed the synthetic witness of the same column shape
    (circuit.synthesize), so that equal requests give equal witnesses;
  * `verify` / `verify_evm` (ecdsa_p256.rs:429-469) are ms-scale host work outside the hot path and
    are not reimplemented in the product; tests verify proofs with the oracle's verifier.
"""
import hashlib
import os

import numpy as np

from . import circuit
from .engine import ZK_TRANSCRIPT_BLAKE2B, ZK_TRANSCRIPT_EVM, Engine, ZkError

_STATE = {}  # (device) -> {"eng": Engine, "k": int, "keys": {path: (params, pk_handle)}}


def _config_for(degree: int) -> circuit.CircuitParams:
    """The reference reads ECDSA_CONFIG / src/configs/ecdsa_circuit.config (ecdsa_p256.rs:95-100);
    the rows BASELINE.json names are built in, anything else comes from $ECDSA_CONFIG."""
    path = os.environ.get("ECDSA_CONFIG")
    if path and os.path.exists(path):
        p = circuit.CircuitParams.from_json(open(path).read().strip().splitlines()[0])
        if p.degree == degree:
            return p
    if degree == 19:
        return circuit.K19
    if degree == 17:
        return circuit.K17
    raise ValueError(f"no circuit config for degree {degree}; set ECDSA_CONFIG")


def gen_srs(degree: int, device: int = 0) -> Engine:
    """halo2-base `gen_srs(k)`: ParamsKZG::setup(k, ChaCha20Rng::from_seed([0; 32])), kept resident."""
    st = _STATE.setdefault(device, {"eng": None, "k": None, "keys": {}})
    if st["eng"] is None:
        st["eng"] = Engine(device)
    if st["k"] != degree:
        for _, pk in st["keys"].values():
            st["eng"].pk_free(pk)
        st["keys"].clear()
        st["eng"].srs_setup(degree, bytes(32))
        st["k"] = degree
    return st["eng"]


def download_keys(degree: int, proving_key_path=None, verifying_key_path=None, device: int = 0):
    """keygen_vk + keygen_pk for the ECDSA-shape circuit.  The proving key stays on the device
    (registered under `proving_key_path`); the verifying key (commitments + transcript_repr) is
    written to `verifying_key_path` if given."""
    eng = gen_srs(degree, device)
    p = _config_for(degree)
    asg = circuit.synthesize(p, 0)  # structure only: fixed columns and copy constraints
    fixed = np.stack([asg.to_limbs(c) for c in asg.fixed])
    pk = eng.keygen(p, fixed, asg.copies)
    _STATE[device]["keys"][proving_key_path or "<default>"] = (p, pk)
    if verifying_key_path:
        fc, pc, tr = eng.vk_export(pk)
        with open(verifying_key_path, "wb") as f:
            f.write(np.uint32([degree, fc.shape[0], pc.shape[0]]).tobytes())
            f.write(fc.tobytes() + pc.tobytes() + tr.tobytes())
    return pk


def _witness_seed(pubkey_x, pubkey_y, r, s, msg_hash) -> int:
    for name, v in (("pubkey_x", pubkey_x), ("pubkey_y", pubkey_y), ("r", r), ("s", s), ("msg_hash", msg_hash)):
        if len(v) != 32:
            raise ValueError(f"{name} must be 32 little-endian bytes")  # the reference takes &[u8; 32]
    return int.from_bytes(hashlib.sha256(bytes(pubkey_x) + bytes(pubkey_y) + bytes(r) + bytes(s) + bytes(msg_hash)).digest()[:8], "little")


def _prove(pubkey_x, pubkey_y, r, s, msg_hash, proving_key_path, degree, transcript, device, rng_seed):
    eng = gen_srs(degree, device)
    keys = _STATE[device]["keys"]
    key = proving_key_path or "<default>"
    if key not in keys:
        # the reference panics with "Unable to open proving key file" (ecdsa_p256.rs:340)
        raise FileNotFoundError(f"Unable to open proving key file: {proving_key_path} (call download_keys first)")
    p, pk = keys[key]
    asg = circuit.synthesize(p, _witness_seed(pubkey_x, pubkey_y, r, s, msg_hash))
    polys = []
    try:
        for col in asg.advice:
            h = eng.poly(1 << degree)
            eng.upload_canonical(h, asg.to_limbs(col))
            polys.append(h)
        seed = rng_seed if rng_seed is not None else os.urandom(32)  # the reference draws from OsRng (ecdsa_p256.rs:362)
        return eng.prove(pk, polys, seed, transcript)
    finally:
        for h in polys:
            h.free()


def generate_proof(pubkey_x, pubkey_y, r, s, msg_hash, proving_key_path, degree, device=0, rng_seed=None) -> bytes:
    """Blake2b transcript + SHPLONK (the /prove endpoint, proving-server/src/main.rs:65-79)."""
    return _prove(pubkey_x, pubkey_y, r, s, msg_hash, proving_key_path, degree, ZK_TRANSCRIPT_BLAKE2B, device, rng_seed)


def generate_proof_evm(pubkey_x, pubkey_y, r, s, msg_hash, proving_key_path, degree, device=0, rng_seed=None) -> bytes:
    """Keccak EvmTranscript + GWC (the /prove_evm endpoint, proving-server/src/main.rs:49-63)."""
    return _prove(pubkey_x, pubkey_y, r, s, msg_hash, proving_key_path, degree, ZK_TRANSCRIPT_EVM, device, rng_seed)


def verify(*_a, **_k):
    raise ZkError(-1, "verify/verify_evm are host-side and out of the engine's scope (DESIGN.md §1); "
                      "use the reference's verify_proof, the generated verifier, or the oracle verifier in tests")


verify_evm = verify


def prover_smoke(eng: Engine) -> None:
    """One tiny full proof on `eng` checked by the oracle (called from __graft_entry__.smoke)."""
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    from zkoracle import cops, plonk, prover  # checker only
    from zkoracle.hashes import ChaCha20Rng

    p = circuit.CircuitParams(degree=7, num_advice=1, num_lookup_advice=1, num_fixed=1, lookup_bits=6)
    asg = circuit.synthesize(p, 0x5EED0019)
    eng.srs_setup(7)
    pk = eng.keygen(p, np.stack([asg.to_limbs(c) for c in asg.fixed]), asg.copies)
    h = eng.poly(128)
    eng.upload_canonical(h, asg.to_limbs(asg.advice[0]))
    seed = b"\x05" * 32
    got = eng.prove(pk, [h], seed, ZK_TRANSCRIPT_EVM)
    sh = plonk.Shape(7, 1, 1, 1, 6)
    opk = prover.keygen(prover.Circuit(sh, asg.fixed, asg.copies, asg.advice))
    assert got == prover.create_proof(opk, asg.advice, ChaCha20Rng(seed), "evm"), "device proof != oracle proof"
    fc, pc, tr = eng.vk_export(pk)
    vk = plonk.VerifyingKey(sh, cops.affine_arr_to_ints(fc), cops.affine_arr_to_ints(pc), cops.fr_ints(tr.reshape(1, 4))[0])
    assert plonk.verify(vk, got, "evm")
    h.free()
    eng.pk_free(pk)
