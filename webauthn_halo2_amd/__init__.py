"""Importable alias of the `webauthn-halo2_amd/` package directory (a hyphen is not a valid Python identifier): the
package's modules live there; this file only points the import system at them and names the public surface."""
import os as _os

__path__.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "webauthn-halo2_amd"))
from .engine import Engine, ZkError, lib_path, load_library  # noqa: F401,E402
from . import batch, circuit, ecdsa_p256, proving_server  # noqa: F401,E402

__all__ = ["Engine", "ZkError", "lib_path", "load_library", "circuit", "batch", "ecdsa_p256", "proving_server"]
