"""Public surface of the package (re-exported by `webauthn_halo2_amd`)."""
from .engine import Engine, ZkError, lib_path, load_library  # noqa: F401
from . import batch, circuit, ecdsa_p256, proving_server  # noqa: F401

__all__ = ["Engine", "ZkError", "lib_path", "load_library", "circuit", "batch", "ecdsa_p256", "proving_server"]
