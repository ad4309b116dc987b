"""JSON-in / hex-out request contract of the reference's proving server, on the resident engine.

Mirrors the two proving endpoints of proving-server/src/main.rs (the Rocket server itself is out of scope —
SURVEY.md §2 #9 — only its request/response contract is reproduced so that a batch of recorded requests can
be replayed against the engine):

    struct ProveRequestBody { r, s, pubkey_x, pubkey_y, msghash: [u8; 32], proving_key_path: String }
                                                                       proving-server/src/main.rs:39-47
    POST /prove_evm  -> hex::encode(generate_proof_evm(..., DEGREE))   main.rs:49-63
    POST /prove      -> hex::encode(generate_proof(..., DEGREE))       main.rs:65-79
    POST /setup      -> download_keys(DEGREE, "./keys/proving_key.pk", "./keys/verifying_key.vk")   main.rs:29-37
    const DEGREE: u32 = 17                                             main.rs:17

The five byte arrays are LITTLE-endian, as the web client builds them (web-demo/src/pages/index.tsx:285-293:
big-endian WebAuthn values reversed before the POST).  serde rejects a body whose arrays are not exactly 32
integers in 0..=255; so does `parse_request`.

The proofs are those of `ecdsa_p256.generate_proof*_synthetic` (the same-shape synthetic circuit, ES256
signature checked on the host — see that module's docstring for what that does and does not mean).
"""
import json
import threading

from . import ecdsa_p256

DEGREE = 17  # proving-server/src/main.rs:17
FIELDS = ("r", "s", "pubkey_x", "pubkey_y", "msghash")


def parse_request(body):
    """ProveRequestBody from a JSON string / dict: five [u8; 32] arrays + proving_key_path."""
    if isinstance(body, (str, bytes, bytearray)):
        body = json.loads(body)
    if not isinstance(body, dict):
        raise ValueError("request body must be a JSON object")
    out = {}
    for f in FIELDS:
        v = body.get(f)
        if not isinstance(v, list) or len(v) != 32 or not all(isinstance(b, int) and not isinstance(b, bool) and 0 <= b <= 255 for b in v):
            raise ValueError(f"{f}: expected an array of 32 integers in 0..=255")  # serde: invalid length / invalid value
        out[f] = bytes(v)
    path = body.get("proving_key_path")
    if not isinstance(path, str):
        raise ValueError("proving_key_path: expected a string")
    out["proving_key_path"] = path
    return out


def setup(device=0, degree=DEGREE, proving_key_path="./keys/proving_key.pk", verifying_key_path=None):
    """POST /setup (and the server's start-up keygen, main.rs:451-456).  The proving key stays resident on
    `device`, registered under `proving_key_path` (the name later requests carry); the verifying key is
    written only when a path is given (the reference writes ./keys/verifying_key.vk)."""
    ecdsa_p256.download_keys(degree, proving_key_path, verifying_key_path, device)
    return "Done"


def _prove(body, evm, device, degree, rng_seed):
    q = parse_request(body)
    fn = ecdsa_p256.generate_proof_evm_synthetic if evm else ecdsa_p256.generate_proof_synthetic
    proof = fn(q["pubkey_x"], q["pubkey_y"], q["r"], q["s"], q["msghash"], q["proving_key_path"], degree, device, rng_seed)
    return proof.hex()  # hex::encode: lowercase, no prefix


def prove_evm(body, device=0, degree=DEGREE, rng_seed=None) -> str:
    """POST /prove_evm: Keccak EvmTranscript + GWC; the hex string the web client puts into userOp.signature."""
    return _prove(body, True, device, degree, rng_seed)


def prove(body, device=0, degree=DEGREE, rng_seed=None) -> str:
    """POST /prove: Blake2b + SHPLONK."""
    return _prove(body, False, device, degree, rng_seed)


def prove_batch(bodies, evm=True, devices=(0,), degree=DEGREE):
    """A recorded batch of requests over several GPUs: request i goes to devices[i % len(devices)], and every device proves
    `ecdsa_p256.PIPELINES_PER_DEVICE` of its requests side by side — one host thread per request in flight (the reference:
    one Rocket worker thread per request, main.rs:457-472).  Every device must have been `setup`.  Returns the hex proofs in
    request order; a failed request yields its exception."""
    bodies = list(bodies)
    out = [None] * len(bodies)
    per = max(1, ecdsa_p256.PIPELINES_PER_DEVICE)
    workers = len(devices) * per

    def work(q):
        for i in range(q, len(bodies), workers):
            try:
                out[i] = _prove(bodies[i], evm, devices[q % len(devices)], degree, None)
            except Exception as e:  # the reference answers 500 for that request and keeps serving
                out[i] = e

    ths = [threading.Thread(target=work, args=(q,)) for q in range(workers)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    return out
