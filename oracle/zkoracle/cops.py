"""ctypes access to oracle/liboracle.so (C restatement of MSM / NTT / field ops).

Oracle (test infrastructure) — see oracle/c/oracle.c.  Arrays are numpy uint64
with the Rust memory image (4 LE limbs per field element, Montgomery form).
"""
import ctypes
import os
import subprocess

import numpy as np

from .field import P, R, MONT_R

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(os.path.dirname(_HERE), "liboracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(_HERE)])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        u64p = ctypes.POINTER(ctypes.c_uint64)
        _lib.orc_msm_bn254.argtypes = [u64p, u64p, ctypes.c_size_t, u64p, ctypes.c_int]
        _lib.orc_ntt_bn254_fr.argtypes = [u64p, u64p, ctypes.c_uint32, ctypes.c_int]
        _lib.orc_fixed_base_g1.argtypes = [u64p, ctypes.c_size_t, u64p, ctypes.c_int]
        _lib.orc_fe_mul.argtypes = [ctypes.c_int, u64p, u64p, u64p]
        _lib.orc_fe_to_mont.argtypes = [ctypes.c_int, u64p, u64p, ctypes.c_size_t]
        _lib.orc_fe_from_mont.argtypes = [ctypes.c_int, u64p, u64p, ctypes.c_size_t]
        _lib.orc_g1_to_affine.argtypes = [u64p, u64p]
        _lib.orc_fr_powers.argtypes = [u64p, u64p, u64p, ctypes.c_size_t]
    return _lib


def ptr(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))


def ncpu():
    return os.cpu_count() or 1


# ---- int <-> limb-array conversions ----------------------------------------

def ints_to_arr(xs):
    """list of ints (< 2^256) -> (n,4) uint64 array (no Montgomery conversion)."""
    b = b"".join(int(x).to_bytes(32, "little") for x in xs)
    return np.frombuffer(b, dtype=np.uint64).reshape(-1, 4).copy()


def arr_to_ints(a):
    b = np.ascontiguousarray(a, dtype=np.uint64).tobytes()
    return [int.from_bytes(b[i:i + 32], "little") for i in range(0, len(b), 32)]


def to_mont_arr(a, which=0):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    out = np.empty_like(a)
    lib().orc_fe_to_mont(which, ptr(a), ptr(out), a.size // 4)
    return out


def from_mont_arr(a, which=0):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    out = np.empty_like(a)
    lib().orc_fe_from_mont(which, ptr(a), ptr(out), a.size // 4)
    return out


def fr_mont(xs):
    return to_mont_arr(ints_to_arr(xs), 0)


def fr_ints(a):
    return arr_to_ints(from_mont_arr(a, 0))


# ---- operators ---------------------------------------------------------------

def msm(scalars_mont, bases_mont, nthreads=None):
    """best_multiexp restatement -> Jacobian Montgomery (12 limbs)."""
    n = scalars_mont.shape[0]
    out = np.zeros(12, dtype=np.uint64)
    lib().orc_msm_bn254(ptr(scalars_mont), ptr(bases_mont), n, ptr(out), nthreads or ncpu())
    return out


def jac_to_affine_ints(jac):
    aff = np.zeros(8, dtype=np.uint64)
    lib().orc_g1_to_affine(ptr(np.ascontiguousarray(jac)), ptr(aff))
    x, y = arr_to_ints(from_mont_arr(aff.reshape(2, 4), 1))
    return None if (x == 0 and y == 0) else (x, y)


def affine_arr_to_ints(aff):
    v = arr_to_ints(from_mont_arr(np.ascontiguousarray(aff).reshape(-1, 4), 1))
    return [None if (v[i] == 0 and v[i + 1] == 0) else (v[i], v[i + 1]) for i in range(0, len(v), 2)]


def ntt(a_mont, omega_int, log_n, nthreads=None):
    """best_fft restatement, in place on a copy; returns the array."""
    a = np.ascontiguousarray(a_mont, dtype=np.uint64).copy()
    w = fr_mont([omega_int])
    lib().orc_ntt_bn254_fr(ptr(a), ptr(w), log_n, nthreads or ncpu())
    return a


def fixed_base_g1(scalars_mont, nthreads=None):
    n = scalars_mont.shape[0]
    out = np.zeros((n, 8), dtype=np.uint64)
    lib().orc_fixed_base_g1(ptr(scalars_mont), n, ptr(out), nthreads or ncpu())
    return out


def fr_powers(base_int, n, first_int=1):
    out = np.zeros((n, 4), dtype=np.uint64)
    lib().orc_fr_powers(ptr(fr_mont([base_int])), ptr(fr_mont([first_int])), ptr(out), n)
    return out
