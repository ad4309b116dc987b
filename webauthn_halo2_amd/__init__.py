"""Importable alias of the `webauthn-halo2_amd/` package directory (a hyphen is
not a valid Python identifier).  All code lives in `webauthn-halo2_amd/`."""
import os as _os

__path__.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "webauthn-halo2_amd"))
from ._pkg import *  # noqa: F401,F403,E402
from ._pkg import __all__  # noqa: E402
