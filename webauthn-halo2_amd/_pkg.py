"""Public surface of the package (re-exported by `webauthn_halo2_amd`)."""
from .engine import Engine, ZkError, lib_path, load_library  # noqa: F401
from . import batch, circuit, ecdsa_p256  # noqa: F401
from .ecdsa_p256 import prover_smoke  # noqa: F401

__all__ = ["Engine", "ZkError", "lib_path", "load_library", "circuit", "batch", "ecdsa_p256", "prover_smoke"]
