"""Host-side mirror of the reference's proving API (halo2-circuits/src/ecc/ecdsa_p256.rs) on top of
the resident engine.

    download_keys(degree, proving_key_path, verifying_key_path)        ecdsa_p256.rs:256-272
    create_proof_from_advice(advice_columns, ..., transcript)          the engine's real input: the advice
                                                                       columns `ECDSACircuit::synthesize`
                                                                       (ecdsa_p256.rs:117-206) leaves behind
    generate_proof_synthetic / generate_proof_evm_synthetic            request-shaped stand-ins for
                                                                       generate_proof (:379-427) / _evm (:329-377)

What the scope of this repository imposes, stated where a caller will read it:

  * The reference re-reads the SRS and the proving key from disk on EVERY request
    (ecdsa_p256.rs:338-343); here `gen_srs` and the key stay resident on the device
     — the path arguments select a cached, resident key.
  * The secp256r1 witness generation (`ECDSACircuit::synthesize`) needs the Rust halo2-ecc chips and stays
    on the host side of the real integration.  The product entry point is therefore
    `create_proof_from_advice`: it takes advice columns and proves them.
  * The request-shaped functions carry `_synthetic` in their name because the circuit they prove is the
    SAME-SHAPE SYNTHETIC circuit of `circuit.synthesize`, not the secp256r1 verification circuit: a proof
    from them says nothing about the signature inside the SNARK.  They do check the ES256 signature on the
    host first (plain secp256r1 ECDSA verification of (r, s) over msg_hash under the public key) and refuse
    an invalid request, so that, unlike a bare seed-hash, an invalid signature or arbitrary bytes never
    yields a verifying proof.  The reference's `generate_proof*` names are deliberately NOT exported.
  * `verify` / `verify_evm` (ecdsa_p256.rs:429-469) are ms-scale host work outside the hot path and
    are not reimplemented in the product; tests verify proofs with the oracle's verifier.
"""
import hashlib
import os
import queue
import threading

import numpy as np

from . import circuit
from .engine import ZK_TRANSCRIPT_BLAKE2B, ZK_TRANSCRIPT_EVM, Engine

# (device) -> {"eng": Engine, "k": int, "keys": {path: (params, pk_handle)}, "slots": {columns: [[Poly]]},
#              "extra": [{"eng": Engine sharing the first one's SRS, "keys": {path: pk_handle}, "slots": {..}}], "free": Queue of pipeline indices}
_STATE = {}
_SLOTS_LOCK = threading.Lock()
_STATE_LOCK = threading.RLock()  # set-up / tear-down of a device's resident state (gen_srs with a new degree, shutdown)
# Proof pipelines per device = requests proved CONCURRENTLY on it (the reference: one Rocket worker thread per request,
# proving-server/src/main.rs:457-472).  Each is a context of its own (own key, own workspace, shared SRS and window tables).
# Four is the measured optimum for the server's default shape since round 5: 143 / 184 / 208 / 224 / 208 proofs/s at k = 17 with
# 1 / 2 / 3 / 4 / 6 (tools/inflight_k17.py, profiles/r5_k17_inflight.txt; round 4: 144 / 195 / 173 / 183), and what k = 19 batches
# want too (batch.py, bench.py).  A request that arrives alone still proves alone: pipeline 0 is taken first.
PIPELINES_PER_DEVICE = 4
_MAX_SLOT_SETS = 4  # parked request-slot sets per column count: the server's usual number of requests in flight per device
# Resident bytes per pipeline by degree (tools/mem_footprint.py: SRS + window tables are shared, every pipeline keeps its own
# key, prover workspace and three MSM lanes), plus the request-slot sets it may park (_MAX_SLOT_SETS advice sets of 32 B x n)
_PIPELINE_BYTES = {17: 2.0 * 2**30, 18: 3.2 * 2**30, 19: 5.3 * 2**30, 20: 10.5 * 2**30, 21: 21.0 * 2**30}


def pipelines_for(degree: int, device: int = 0) -> int:
    """How many pipelines a device gets at this degree: PIPELINES_PER_DEVICE, fewer when they would not leave a quarter of the
    device's FREE memory to everything else (another process's contexts, lock-step members: 1.4 GiB each at k = 19)."""
    from .engine import device_mem_info

    want = max(1, PIPELINES_PER_DEVICE)
    per = _PIPELINE_BYTES.get(degree, _PIPELINE_BYTES[21] * (1 << max(0, degree - 21)) if degree > 21 else _PIPELINE_BYTES[17])
    per += _MAX_SLOT_SETS * 32 * (1 << degree)
    try:
        free, _ = device_mem_info(device)
    except Exception:
        return want
    return max(1, min(want, int(0.75 * free // per)))


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


def _drain(st):
    """Take every pipeline of a device out of its free queue: returns once no request holds one (requests in flight finish
    first — their `finally` puts the index back into THIS queue object), so contexts can be closed without a use after close."""
    q = st.get("free")
    if q is None:
        return
    for _ in range(1 + len(st["extra"])):
        q.get()
    st["free"] = None


class _Hold:
    """Every pipeline of a device taken out of its free queue for the duration of a `with` block (requests in flight finish
    first, new ones wait): a key can be freed and replaced without a request proving under it.  The indices go back into the
    same queue object, pipeline 0 on top."""

    def __init__(self, st):
        self.q, self.n = st.get("free"), 1 + len(st["extra"])

    def __enter__(self):
        if self.q is not None:
            for _ in range(self.n):
                self.q.get()
        return self

    def __exit__(self, *exc):
        if self.q is not None:
            for i in range(self.n - 1, -1, -1):
                self.q.put(i)
        return False


def gen_srs(degree: int, device: int = 0) -> Engine:
    """halo2-base `gen_srs(k)`: ParamsKZG::setup(k, ChaCha20Rng::from_seed([0; 32])), kept resident."""
    with _STATE_LOCK:
        return _gen_srs_locked(degree, device)


def _gen_srs_locked(degree, device):
    st = _STATE.setdefault(device, {"eng": None, "k": None, "keys": {}, "slots": {}, "extra": [], "free": None})
    if st["eng"] is None:
        st["eng"] = Engine(device)
    if st["k"] != degree:
        _drain(st)  # requests still proving under the old SRS hold a pipeline: wait for them before anything is closed
        for _, pk in st["keys"].values():
            st["eng"].pk_free(pk)
        st["keys"].clear()
        with _SLOTS_LOCK:
            for sets in st["slots"].values():
                for polys in sets:
                    for h in polys:
                        h.free()
            st["slots"].clear()
        for m in st["extra"]:  # the further pipelines saw the old SRS: they go with it
            m["eng"].close()
        st["extra"] = []
        st["eng"].srs_setup(degree, bytes(32))
        st["k"] = degree
        st["extra"] = [{"eng": Engine(device, share_with=st["eng"]), "keys": {}, "slots": {}} for _ in range(pipelines_for(degree, device) - 1)]
        st["free"] = queue.LifoQueue()
        for i in range(len(st["extra"]), -1, -1):  # pipeline 0 (the first context) on top: a lone request takes it
            st["free"].put(i)
    return st["eng"]


def shutdown(device=None):
    """Release the resident state (every pipeline's context, keys and request slots) of `device`, or of all devices."""
    with _STATE_LOCK:
        for d in ([device] if device is not None else list(_STATE)):
            st = _STATE.get(d)
            if not st:
                continue
            _drain(st)  # (a request that arrives from now on finds no queue: "no resident state")
            _STATE.pop(d, None)
            st["k"] = None
            for m in st["extra"][::-1]:
                m["eng"].close()
            if st["eng"] is not None:
                st["eng"].close()


def _pipeline(st, i):
    """(engine, {key name: pk handle}, request slots) of pipeline i of a device: 0 is the first context."""
    if i == 0:
        return st["eng"], {name: v[1] for name, v in st["keys"].items()}, st["slots"]
    m = st["extra"][i - 1]
    return m["eng"], m["keys"], m["slots"]


def download_keys(degree: int, proving_key_path=None, verifying_key_path=None, device: int = 0):
    """keygen_vk + keygen_pk for the ECDSA-shape circuit.  The proving key stays on the device
    (registered under `proving_key_path`; a key already registered under that name is freed first); the
    verifying key is written to `verifying_key_path` if given, as the reference writes it
    (`vk.to_bytes(SerdeFormat::RawBytes)`, ecdsa_p256.rs:266-270: the VerifyingKey::write image of zk_vk_write)."""
    p = _config_for(degree)
    asg = circuit.synthesize(p, 0)  # structure only: fixed columns and copy constraints
    fixed = np.stack([asg.to_limbs(c) for c in asg.fixed])
    with _STATE_LOCK:  # (one set-up / tear-down of a device's state at a time)
        eng = _gen_srs_locked(degree, device)
        st = _STATE[device]
        keys = st["keys"]
        name = proving_key_path or "<default>"
        # /setup called again while requests are proving: they finish under the old key first — the pipelines are held while the
        # resident key (GBs at k = 17 / 19) is freed and replaced, not leaked and not pulled from under a proof
        with _Hold(st):
            if name in keys:
                eng.pk_free(keys.pop(name)[1])
            pk = eng.keygen(p, fixed, asg.copies)
            keys[name] = (p, pk)
            for m in st["extra"]:  # every further pipeline of the device holds the key too (its own workspace comes with it)
                if name in m["keys"]:
                    m["eng"].pk_free(m["keys"].pop(name))
                m["keys"][name] = m["eng"].keygen(p, fixed, asg.copies)
        if verifying_key_path:
            with open(verifying_key_path, "wb") as f:
                f.write(eng.vk_write(pk).tobytes())
    return pk


def _resident_key(proving_key_path, degree, device):
    eng = gen_srs(degree, device)
    keys = _STATE[device]["keys"]
    key = proving_key_path or "<default>"
    if key not in keys:
        # the reference panics with "Unable to open proving key file" (ecdsa_p256.rs:340)
        raise FileNotFoundError(f"Unable to open proving key file: {proving_key_path} (call download_keys first)")
    return (eng,) + keys[key]


def create_proof_from_advice(advice_columns, proving_key_path, degree, transcript=ZK_TRANSCRIPT_BLAKE2B, device=0,
                             rng_seed=None) -> bytes:
    """create_proof over host-synthesized advice columns — what an unchanged Rust host hands the engine
    after `ECDSACircuit::synthesize`.  `advice_columns`: sequence of (n, 4) uint64 arrays of canonical
    little-endian limbs, one per advice column of the key's shape."""
    _resident_key(proving_key_path, degree, device)  # (raises for an unknown key)
    n = 1 << degree
    cols = []
    for col in advice_columns:
        col = np.ascontiguousarray(col, dtype=np.uint64)
        if col.shape != (n, 4):
            raise ValueError("an advice column must be an (n, 4) array of canonical limbs")
        cols.append(col)
    # a pipeline of the device for this request: concurrent requests prove side by side (PIPELINES_PER_DEVICE), further ones wait
    st = _STATE[device]
    q = st["free"]  # THIS queue object gets the index back, whatever happens to the device's state meanwhile (_drain waits on it)
    if q is None or st["k"] != degree:
        raise RuntimeError("the device's resident state was released or replaced while this request was being set up")
    while True:
        try:
            which = q.get(timeout=0.05)
            break
        except queue.Empty:  # every pipeline is busy — or the device's state went away while we waited (its queue is drained for good)
            if st["free"] is not q:
                raise RuntimeError("the device's resident state was released or replaced while this request was waiting")
    try:
        eng, pks, slots = _pipeline(st, which)
        name = proving_key_path or "<default>"
        if st["k"] != degree or name not in pks:  # gen_srs(another degree) ran between the key lookup above and here
            raise FileNotFoundError(f"Unable to open proving key file: {proving_key_path} (the resident key was replaced)")
        return _prove_on(st, eng, pks[name], slots, cols, n, degree, transcript, rng_seed)
    finally:
        q.put(which)


def _prove_on(st, eng, pk, slots, cols, n, degree, transcript, rng_seed):
    # request slots: the columns' device buffers are kept between requests (a hipFree per request would wait for the whole
    # device, i.e. for every other request in flight on it); concurrent requests each take a set of their own
    with _SLOTS_LOCK:
        free = slots.setdefault(len(cols), [])
        polys = free.pop() if free else None
    if polys is None:
        polys = []
        try:
            for _ in cols:
                polys.append(eng.poly(n))
        except Exception:  # a partial allocation must not leak the handles already made
            for h in polys:
                h.free()
            raise
    try:
        for h, col in zip(polys, cols):
            eng.upload_canonical(h, col)
        seed = rng_seed if rng_seed is not None else os.urandom(32)  # the reference draws from OsRng (ecdsa_p256.rs:362)
        return eng.prove(pk, polys, seed, transcript)
    finally:
        with _SLOTS_LOCK:
            keep = slots.setdefault(len(cols), [])
            # at most _MAX_SLOT_SETS parked sets per column count (a set is GBs at k = 19 with many columns): more
            # concurrent requests than that allocate and free their own
            if st["k"] == degree and len(keep) < _MAX_SLOT_SETS:  # (`st` as captured at entry: shutdown() may have dropped _STATE[device])
                keep.append(polys)
            else:  # the SRS was replaced meanwhile (these buffers belong to the old size), or enough sets are parked
                for h in polys:
                    h.free()


# ---- ES256 (secp256r1 ECDSA) request validation, host side -------------------------------------------
_P = 0xFFFFFFFF00000001000000000000000000000000FFFFFFFFFFFFFFFFFFFFFFFF
_N = 0xFFFFFFFF00000000FFFFFFFFFFFFFFFFBCE6FAADA7179E84F3B9CAC2FC632551
_B = 0x5AC635D8AA3A93E7B3EBBD55769886BC651D06B0CC53B0F63BCE3C3E27D2604B
_G = (0x6B17D1F2E12C4247F8BCE6E563A440F277037D812DEB33A0F4A13945D898C296,
      0x4FE342E2FE1A7F9B8EE7EB4A7C0F9E162BCE33576B315ECECBB6406837BF51F5)


def _p256_add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    if a[0] == b[0]:
        if (a[1] + b[1]) % _P == 0:
            return None
        lam = 3 * (a[0] * a[0] - 1) * pow(2 * a[1], -1, _P) % _P  # a = -3
    else:
        lam = (b[1] - a[1]) * pow(b[0] - a[0], -1, _P) % _P
    x = (lam * lam - a[0] - b[0]) % _P
    return x, (lam * (a[0] - x) - a[1]) % _P


def _p256_mul(k, pt):
    acc = None
    while k:
        if k & 1:
            acc = _p256_add(acc, pt)
        pt = _p256_add(pt, pt)
        k >>= 1
    return acc


def es256_verify(pubkey_x: bytes, pubkey_y: bytes, r: bytes, s: bytes, msg_hash: bytes) -> bool:
    """Plain secp256r1 ECDSA verification of the request, all five fields 32 little-endian bytes as the web
    client posts them (web-demo/src/pages/index.tsx:285-293; `Fp::from_bytes` / `Fq::from_bytes` at
    ecdsa_p256.rs:345-352 reject non-canonical encodings — so does this)."""
    x, y = int.from_bytes(pubkey_x, "little"), int.from_bytes(pubkey_y, "little")
    ri, si, z = int.from_bytes(r, "little"), int.from_bytes(s, "little"), int.from_bytes(msg_hash, "little")
    if x >= _P or y >= _P or z >= _N or not (0 < ri < _N) or not (0 < si < _N):
        return False
    if (y * y - (x * x * x - 3 * x + _B)) % _P:
        return False  # Secp256r1Affine::from_xy is None off the curve
    w = pow(si, -1, _N)
    pt = _p256_add(_p256_mul(z * w % _N, _G), _p256_mul(ri * w % _N, (x, y)))
    return pt is not None and pt[0] % _N == ri


def _witness_seed(pubkey_x, pubkey_y, r, s, msg_hash) -> int:
    return int.from_bytes(hashlib.sha256(bytes(pubkey_x) + bytes(pubkey_y) + bytes(r) + bytes(s) + bytes(msg_hash)).digest()[:8], "little")


def _prove_synthetic(pubkey_x, pubkey_y, r, s, msg_hash, proving_key_path, degree, transcript, device, rng_seed):
    for name, v in (("pubkey_x", pubkey_x), ("pubkey_y", pubkey_y), ("r", r), ("s", s), ("msg_hash", msg_hash)):
        if len(v) != 32:
            raise ValueError(f"{name} must be 32 little-endian bytes")  # the reference takes &[u8; 32]
    if not es256_verify(pubkey_x, pubkey_y, r, s, msg_hash):
        # the real circuit would be unsatisfiable; never let such a request come back with a verifying proof
        raise ValueError("invalid ES256 signature (or non-canonical field encoding): request refused")
    _, p, _ = _resident_key(proving_key_path, degree, device)
    asg = circuit.synthesize(p, _witness_seed(pubkey_x, pubkey_y, r, s, msg_hash))
    return create_proof_from_advice([asg.to_limbs(col) for col in asg.advice], proving_key_path, degree, transcript, device, rng_seed)


def generate_proof_synthetic(pubkey_x, pubkey_y, r, s, msg_hash, proving_key_path, degree, device=0, rng_seed=None) -> bytes:
    """Request shape of `generate_proof` (Blake2b + SHPLONK, the /prove endpoint, proving-server/src/main.rs:65-79)
    over the SYNTHETIC same-shape circuit — see the module docstring."""
    return _prove_synthetic(pubkey_x, pubkey_y, r, s, msg_hash, proving_key_path, degree, ZK_TRANSCRIPT_BLAKE2B, device, rng_seed)


def generate_proof_evm_synthetic(pubkey_x, pubkey_y, r, s, msg_hash, proving_key_path, degree, device=0, rng_seed=None) -> bytes:
    """Request shape of `generate_proof_evm` (Keccak EvmTranscript + GWC, the /prove_evm endpoint,
    proving-server/src/main.rs:49-63) over the SYNTHETIC same-shape circuit — see the module docstring."""
    return _prove_synthetic(pubkey_x, pubkey_y, r, s, msg_hash, proving_key_path, degree, ZK_TRANSCRIPT_EVM, device, rng_seed)


# `verify` / `verify_evm` (ecdsa_p256.rs:429-469) are not mirrored: verification is host-side pairing work outside the engine's
# path — the reference's verify_proof, its generated verifier, or (in tests) the oracle verifier check the proofs.
