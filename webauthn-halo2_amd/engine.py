"""ctypes binding of libzkmi355.so (include/zkmi355.h) — the only way Python
reaches the engine.  There is no CPU fallback: a missing library or a missing
gfx950 device raises `ZkError`.

Arrays are numpy uint64 with the Rust memory images (Fr/Fq: 4 LE limbs,
Montgomery; G1Affine: 8 limbs; G1 Jacobian: 12 limbs).
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

ZK_BASIS_MONOMIAL = 0
ZK_BASIS_LAGRANGE = 1
ZK_T_MSM, ZK_T_NTT, ZK_T_QUOTIENT, ZK_T_EVAL, ZK_T_MSM_ACCUM, ZK_T_MSM_COLUMNS, ZK_T_MSM_TAIL_MAIN, ZK_T_MSM_TAIL = 0, 1, 2, 3, 4, 5, 6, 7


ZK_TRANSCRIPT_BLAKE2B, ZK_TRANSCRIPT_EVM = 0, 1
ZK_SERDE_PROCESSED, ZK_SERDE_RAW_BYTES, ZK_SERDE_RAW_BYTES_UNCHECKED = 0, 1, 2
ZK_OPT_MSM_WINDOW, ZK_OPT_MSM_BATCH, ZK_OPT_NTT_MAX_RADIX_LOG2, ZK_OPT_GP_BATCH_INVERT, ZK_OPT_MSM_TAIL_STREAM = 1, 2, 3, 4, 5
ZK_OPT_MSM_TAIL_MAIN_ABOVE, ZK_OPT_BATCH_PASS_COLUMNS, ZK_OPT_XFORM_STREAM, ZK_OPT_MSM_STREAM, ZK_OPT_MSM_T1, ZK_OPT_STREAM_AUDIT = 6, 7, 8, 9, 10, 11
ZK_OPT_STREAM_PRIORITY, ZK_OPT_QUOTIENT_DOMAIN, ZK_OPT_ACTIVITY_HOLD = 12, 13, 14
ZK_SCHEME_DEFAULT, ZK_SCHEME_GWC, ZK_SCHEME_SHPLONK = 0, 1, 2


def device_pci_bus_id(device=0):
    """PCI address of a device, e.g. '0000:c1:00.0' (zk_device_pci_bus_id)."""
    buf = ctypes.create_string_buffer(32)
    rc = load_library().zk_device_pci_bus_id(device, buf, len(buf))
    if rc:
        raise ZkError(rc, "zk_device_pci_bus_id")
    return buf.value.decode().lower()


def device_mem_info(device=0):
    """(free, total) bytes of a device's memory (zk_device_mem_info)."""
    free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
    rc = load_library().zk_device_mem_info(device, ctypes.byref(free), ctypes.byref(total))
    if rc:
        raise ZkError(rc, "zk_device_mem_info")
    return free.value, total.value


class PinnedArray:
    """A numpy array over page-locked host memory (zk_host_alloc): the buffers a host hands to upload / upload_canonical.
    Keep the object alive while `a` is in use; free() (or garbage collection) returns the memory."""

    def __init__(self, shape, dtype=np.uint64):
        self.L = load_library()
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self.ptr = self.L.zk_host_alloc(nbytes)
        if not self.ptr:
            raise ZkError(-2, "zk_host_alloc")
        self.a = np.frombuffer((ctypes.c_uint8 * nbytes).from_address(self.ptr), dtype=dtype).reshape(shape)

    def free(self):
        if self.ptr:
            self.a = None
            self.L.zk_host_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class CircuitParamsC(ctypes.Structure):
    _fields_ = [("k", ctypes.c_uint32), ("num_advice", ctypes.c_uint32), ("num_lookup_advice", ctypes.c_uint32),
                ("num_fixed", ctypes.c_uint32), ("lookup_bits", ctypes.c_uint32),
                ("num_idle_gate_columns", ctypes.c_uint32)]


class ZkError(RuntimeError):
    def __init__(self, code, what, hip=0):
        self.code = code
        self.hip = hip
        super().__init__(f"{what}: zk error {code}" + (f" (hipError_t {hip})" if hip else ""))


def lib_path():
    # ZKMI355_LIB: an alternative build of the same library (A/B tuning variants, tools/ab_variants.sh)
    return os.environ.get("ZKMI355_LIB") or os.path.join(_HERE, "libzkmi355.so")


def load_library():
    """Load libzkmi355.so; raises if it has not been built (./build.sh)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    p = lib_path()
    if not os.path.exists(p):
        raise ZkError(-4, f"{p} is missing — run ./build.sh (no CPU fallback exists)")
    L = ctypes.CDLL(p)
    u64p = ctypes.POINTER(ctypes.c_uint64)
    vp = ctypes.c_void_p
    sz = ctypes.c_size_t
    u32 = ctypes.c_uint32
    sig = {
        "zk_device_count": ([], ctypes.c_int),
        "zk_device_pci_bus_id": ([ctypes.c_int, ctypes.c_char_p, sz], ctypes.c_int),
        "zk_device_mem_info": ([ctypes.c_int, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_size_t)], ctypes.c_int),
        "zk_host_alloc": ([sz], vp),
        "zk_host_free": ([vp], None),
        "zk_ctx_create": ([ctypes.c_int, ctypes.POINTER(vp)], ctypes.c_int),
        "zk_ctx_create_shared": ([vp, ctypes.POINTER(vp)], ctypes.c_int),
        "zk_ctx_destroy": ([vp], None),
        "zk_strerror": ([ctypes.c_int], ctypes.c_char_p),
        "zk_last_hip_error": ([vp], ctypes.c_int),
        "zk_sync": ([vp], ctypes.c_int),
        "zk_ctx_set_option": ([vp, ctypes.c_int, ctypes.c_int64], ctypes.c_int),
        "zk_msm_bn254": ([vp, u64p, u64p, sz, u64p], ctypes.c_int),
        "zk_msm_srs": ([vp, ctypes.c_int, u64p, sz, u64p], ctypes.c_int),
        "zk_ntt_bn254_fr": ([vp, u64p, u64p, u32], ctypes.c_int),
        "zk_srs_setup": ([vp, u32, ctypes.c_char_p], ctypes.c_int),
        "zk_srs_load": ([vp, u32, u64p, u64p], ctypes.c_int),
        "zk_srs_export": ([vp, ctypes.c_int, u64p, sz, sz], ctypes.c_int),
        "zk_srs_k": ([vp], ctypes.c_int),
        "zk_srs_msm_plan": ([vp, ctypes.POINTER(u32), ctypes.POINTER(u32)], ctypes.c_int),
        "zk_poly_alloc": ([vp, sz, ctypes.POINTER(ctypes.c_uint64)], ctypes.c_int),
        "zk_poly_free": ([vp, ctypes.c_uint64], ctypes.c_int),
        "zk_poly_detach": ([vp, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)], ctypes.c_int),
        "zk_poly_attach": ([vp, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)], ctypes.c_int),
        "zk_poly_discard": ([ctypes.c_uint64], ctypes.c_int),
        "zk_poly_len": ([vp, ctypes.c_uint64, ctypes.POINTER(sz)], ctypes.c_int),
        "zk_poly_upload": ([vp, ctypes.c_uint64, u64p, sz], ctypes.c_int),
        "zk_poly_download": ([vp, ctypes.c_uint64, u64p, sz], ctypes.c_int),
        "zk_poly_copy": ([vp, ctypes.c_uint64, ctypes.c_uint64], ctypes.c_int),
        "zk_commit": ([vp, ctypes.c_uint64, ctypes.c_int, u64p], ctypes.c_int),
        "zk_commit_batch": ([vp, ctypes.POINTER(ctypes.c_uint64), sz, ctypes.c_int, u64p], ctypes.c_int),
        "zk_lagrange_to_coeff": ([vp, ctypes.c_uint64], ctypes.c_int),
        "zk_coeff_to_lagrange": ([vp, ctypes.c_uint64], ctypes.c_int),
        "zk_coeff_to_extended": ([vp, ctypes.c_uint64, ctypes.c_uint64], ctypes.c_int),
        "zk_extended_to_coeff": ([vp, ctypes.c_uint64, sz], ctypes.c_int),
        "zk_eval": ([vp, ctypes.c_uint64, u64p, u64p], ctypes.c_int),
        "zk_last_kernel_ms": ([vp, ctypes.c_int, ctypes.POINTER(ctypes.c_float)], ctypes.c_int),
        "zk_kate_division": ([vp, ctypes.c_uint64, u64p, ctypes.c_uint64], ctypes.c_int),
        "zk_timer_reset": ([vp], ctypes.c_int),
        "zk_timer_stats": ([vp, ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64)], ctypes.c_int),
        "zk_clock_probe": ([vp, ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint64)], ctypes.c_int),
        "zk_audit_report": ([vp, ctypes.POINTER(ctypes.c_uint64), ctypes.c_char_p, sz], ctypes.c_int),
        "zk_keygen": ([vp, ctypes.POINTER(CircuitParamsC), u64p, sz, ctypes.POINTER(ctypes.c_uint32), sz,
                       ctypes.POINTER(ctypes.c_uint64)], ctypes.c_int),
        "zk_pk_set_transcript_repr": ([vp, ctypes.c_uint64, u64p], ctypes.c_int),
        "zk_pk_free": ([vp, ctypes.c_uint64], ctypes.c_int),
        "zk_vk_export": ([vp, ctypes.c_uint64, u64p, u64p, u64p, ctypes.POINTER(ctypes.c_uint32)], ctypes.c_int),
        "zk_proof_size": ([vp, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.POINTER(sz)], ctypes.c_int),
        "zk_prove": ([vp, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64), sz, ctypes.c_char_p, ctypes.c_int,
                      ctypes.c_int, ctypes.c_char_p, sz, ctypes.POINTER(sz)], ctypes.c_int),
        "zk_prove_batch": ([vp, ctypes.c_uint64, sz, ctypes.POINTER(ctypes.c_uint64), sz, ctypes.c_char_p, ctypes.c_int,
                            ctypes.c_int, ctypes.c_char_p, sz, ctypes.POINTER(sz)], ctypes.c_int),
        "zk_srs_write": ([vp, ctypes.c_int, vp, sz, ctypes.POINTER(sz)], ctypes.c_int),
        "zk_srs_read": ([vp, vp, sz, ctypes.c_int], ctypes.c_int),
        "zk_srs_set_g2": ([vp, u64p, u64p], ctypes.c_int),
        "zk_vk_write": ([vp, ctypes.c_uint64, ctypes.c_int, vp, sz, ctypes.POINTER(sz)], ctypes.c_int),
        "zk_vk_load": ([vp, ctypes.c_uint64, vp, sz, ctypes.c_int, u64p], ctypes.c_int),
        "zk_pk_write": ([vp, ctypes.c_uint64, ctypes.c_int, vp, sz, ctypes.POINTER(sz)], ctypes.c_int),
        "zk_pk_read": ([vp, ctypes.POINTER(CircuitParamsC), vp, sz, ctypes.c_int, u64p, ctypes.POINTER(ctypes.c_uint64)], ctypes.c_int),
        "zk_pk_shape": ([vp, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint32)], ctypes.c_int),
        "zk_quotient": ([vp, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64), sz, ctypes.POINTER(ctypes.c_uint64), sz,
                         ctypes.POINTER(ctypes.c_uint64), sz, u64p, u64p, u64p, ctypes.c_int, ctypes.c_uint64], ctypes.c_int),
        "zk_poly_upload_canonical": ([vp, ctypes.c_uint64, u64p, sz], ctypes.c_int),
    }
    for name, (args, res) in sig.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = res
    _LIB = L
    return L


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))


def _arr(a, cols):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    if a.size % cols:
        raise ValueError("bad array shape")
    return a.reshape(-1, cols)


class Poly:
    """Handle of a device-resident vector of Fr."""

    def __init__(self, eng, handle, n):
        self.eng, self.h, self.n = eng, handle, n

    def free(self):
        if self.h:
            self.eng._chk(self.eng.L.zk_poly_free(self.eng.ctx, self.h), "zk_poly_free")
            self.h = 0


class Engine:
    """One zk_ctx = one HIP device + stream."""

    def __init__(self, device=0, share_with=None):
        """share_with: another Engine on the same device whose resident SRS (bases + window tables) this one uses too
        (zk_ctx_create_shared) instead of loading its own copy."""
        self.L = load_library()
        ctx = ctypes.c_void_p()
        if share_with is not None:
            rc = self.L.zk_ctx_create_shared(share_with.ctx, ctypes.byref(ctx))
        else:
            rc = self.L.zk_ctx_create(device, ctypes.byref(ctx))
        if rc != 0:
            raise ZkError(rc, "zk_ctx_create: " + self.L.zk_strerror(rc).decode())
        self.ctx = ctx

    def close(self):
        if getattr(self, "ctx", None):
            self.L.zk_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            raise ZkError(rc, what + ": " + self.L.zk_strerror(rc).decode(), self.L.zk_last_hip_error(self.ctx))

    def set_option(self, option, value):
        self._chk(self.L.zk_ctx_set_option(self.ctx, option, value), "zk_ctx_set_option")

    # ---- fine-grained seam ----------------------------------------------------
    def msm_srs(self, scalars_mont, basis):
        """ParamsKZG::commit / commit_lagrange as a Rust host calls it: host scalars, resident basis."""
        s = _arr(scalars_mont, 4)
        out = np.zeros(12, dtype=np.uint64)
        self._chk(self.L.zk_msm_srs(self.ctx, basis, _p(s), s.shape[0], _p(out)), "zk_msm_srs")
        return out

    def msm(self, scalars_mont, bases_mont):
        s = _arr(scalars_mont, 4)
        b = _arr(bases_mont, 8)
        if s.shape[0] != b.shape[0]:
            raise ValueError("scalars / bases length mismatch")
        out = np.zeros(12, dtype=np.uint64)
        self._chk(self.L.zk_msm_bn254(self.ctx, _p(s), _p(b), s.shape[0], _p(out)), "zk_msm_bn254")
        return out

    def ntt(self, a_mont, omega_mont, log_n):
        a = _arr(a_mont, 4).copy()
        if a.shape[0] != (1 << log_n):
            raise ValueError("length != 2^log_n")
        w = np.ascontiguousarray(omega_mont, dtype=np.uint64).reshape(4)
        self._chk(self.L.zk_ntt_bn254_fr(self.ctx, _p(a), _p(w), log_n), "zk_ntt_bn254_fr")
        return a

    # ---- SRS ----------------------------------------------------------------------
    def srs_setup(self, k, seed=bytes(32)):
        if len(seed) != 32:
            raise ValueError("SRS seed must be 32 bytes")
        self._chk(self.L.zk_srs_setup(self.ctx, k, seed), "zk_srs_setup")

    def srs_load(self, k, g, g_lagrange):
        g, gl = _arr(g, 8), _arr(g_lagrange, 8)
        if g.shape[0] != (1 << k) or gl.shape[0] != (1 << k):
            raise ValueError("SRS arrays must hold 2^k points")
        self._chk(self.L.zk_srs_load(self.ctx, k, _p(g), _p(gl)), "zk_srs_load")

    def srs_msm_plan(self):
        c, w = ctypes.c_uint32(), ctypes.c_uint32()
        self._chk(self.L.zk_srs_msm_plan(self.ctx, ctypes.byref(c), ctypes.byref(w)), "zk_srs_msm_plan")
        return c.value, w.value

    def srs_export(self, basis, first, count):
        out = np.zeros((count, 8), dtype=np.uint64)
        self._chk(self.L.zk_srs_export(self.ctx, basis, _p(out), first, count), "zk_srs_export")
        return out

    # ---- the reference's files (ParamsKZG / VerifyingKey / ProvingKey images) ------------
    def _write(self, fn, what, *args):
        ln = ctypes.c_size_t()
        self._chk(fn(self.ctx, *args, None, 0, ctypes.byref(ln)), what)
        buf = np.empty(ln.value, dtype=np.uint8)
        self._chk(fn(self.ctx, *args, buf.ctypes.data, buf.size, ctypes.byref(ln)), what)
        return buf  # numpy uint8 (a k=19 proving key is 768 MiB: no bytes() copy unless the caller wants one)

    @staticmethod
    def _bytes_arg(data):
        a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
        return a, a.ctypes.data, a.size

    def srs_write(self, fmt=ZK_SERDE_RAW_BYTES):
        return self._write(self.L.zk_srs_write, "zk_srs_write", fmt)

    def srs_read(self, data, fmt=ZK_SERDE_RAW_BYTES):
        keep, ptr, n = self._bytes_arg(data)
        self._chk(self.L.zk_srs_read(self.ctx, ptr, n, fmt), "zk_srs_read")

    def srs_set_g2(self, g2, s_g2):
        a, b = (np.ascontiguousarray(v, dtype=np.uint64).reshape(16) for v in (g2, s_g2))
        self._chk(self.L.zk_srs_set_g2(self.ctx, _p(a), _p(b)), "zk_srs_set_g2")

    def vk_write(self, pk, fmt=ZK_SERDE_RAW_BYTES):
        return self._write(self.L.zk_vk_write, "zk_vk_write", pk, fmt)

    def vk_load(self, pk, data, fmt=ZK_SERDE_RAW_BYTES, transcript_repr=None):
        keep, ptr, n = self._bytes_arg(data)
        t = None if transcript_repr is None else _p(np.ascontiguousarray(transcript_repr, dtype=np.uint64).reshape(4))
        self._chk(self.L.zk_vk_load(self.ctx, pk, ptr, n, fmt, t), "zk_vk_load")

    def pk_write(self, pk, fmt=ZK_SERDE_RAW_BYTES):
        return self._write(self.L.zk_pk_write, "zk_pk_write", pk, fmt)

    def pk_read(self, params, data, fmt=ZK_SERDE_RAW_BYTES, transcript_repr=None):
        cp = CircuitParamsC(params.degree, params.num_advice, params.num_lookup_advice, params.num_fixed, params.lookup_bits,
                            getattr(params, "idle_gate_columns", 0))
        keep, ptr, n = self._bytes_arg(data)
        t = None if transcript_repr is None else _p(np.ascontiguousarray(transcript_repr, dtype=np.uint64).reshape(4))
        h = ctypes.c_uint64()
        self._chk(self.L.zk_pk_read(self.ctx, ctypes.byref(cp), ptr, n, fmt, t, ctypes.byref(h)), "zk_pk_read")
        return h.value

    # ---- resident polynomials ---------------------------------------------------
    def poly(self, n, data=None):
        h = ctypes.c_uint64()
        self._chk(self.L.zk_poly_alloc(self.ctx, n, ctypes.byref(h)), "zk_poly_alloc")
        p = Poly(self, h.value, n)
        if data is not None:
            self.upload(p, data)
        return p

    def upload(self, p, data):
        d = _arr(data, 4)
        self._chk(self.L.zk_poly_upload(self.ctx, p.h, _p(d), d.shape[0]), "zk_poly_upload")

    def download(self, p, n=None):
        n = p.n if n is None else n
        out = np.zeros((n, 4), dtype=np.uint64)
        self._chk(self.L.zk_poly_download(self.ctx, p.h, _p(out), n), "zk_poly_download")
        return out

    def copy(self, dst, src):
        self._chk(self.L.zk_poly_copy(self.ctx, dst.h, src.h), "zk_poly_copy")

    def commit(self, p, basis):
        out = np.zeros(8, dtype=np.uint64)
        self._chk(self.L.zk_commit(self.ctx, p.h, basis, _p(out)), "zk_commit")
        return out

    def commit_batch(self, polys, basis):
        hs = (ctypes.c_uint64 * len(polys))(*[p.h for p in polys])
        out = np.zeros((len(polys), 8), dtype=np.uint64)
        self._chk(self.L.zk_commit_batch(self.ctx, hs, len(polys), basis, _p(out)), "zk_commit_batch")
        return out

    def lagrange_to_coeff(self, p):
        self._chk(self.L.zk_lagrange_to_coeff(self.ctx, p.h), "zk_lagrange_to_coeff")

    def coeff_to_lagrange(self, p):
        self._chk(self.L.zk_coeff_to_lagrange(self.ctx, p.h), "zk_coeff_to_lagrange")

    def coeff_to_extended(self, src, dst):
        self._chk(self.L.zk_coeff_to_extended(self.ctx, src.h, dst.h), "zk_coeff_to_extended")

    def extended_to_coeff(self, ext, n_out):
        self._chk(self.L.zk_extended_to_coeff(self.ctx, ext.h, n_out), "zk_extended_to_coeff")

    def eval(self, p, x_mont):
        x = np.ascontiguousarray(x_mont, dtype=np.uint64).reshape(4)
        out = np.zeros(4, dtype=np.uint64)
        self._chk(self.L.zk_eval(self.ctx, p.h, _p(x), _p(out)), "zk_eval")
        return out

    def poly_detach(self, p):
        """Take a resident vector out of this engine for another one on the same device -> (token, n); see poly_attach."""
        t = ctypes.c_uint64()
        self._chk(self.L.zk_poly_detach(self.ctx, p.h, ctypes.byref(t)), "zk_poly_detach")
        p.h = 0
        return t.value, p.n

    def poly_attach(self, detached):
        """Adopt a vector another engine detached (no copy) -> Poly of this engine."""
        token, n = detached
        h = ctypes.c_uint64()
        self._chk(self.L.zk_poly_attach(self.ctx, token, ctypes.byref(h)), "zk_poly_attach")
        return Poly(self, h.value, n)

    def poly_discard(self, detached):
        """Free a detached vector that will not be attached after all (error paths of a staged load)."""
        self._chk(self.L.zk_poly_discard(detached[0]), "zk_poly_discard")

    def kate_division(self, p, z_mont, q=None):
        """arithmetic::kate_division: q = (p - p(z)) / (X - z), same length as p (top coefficient 0); in place by default."""
        z = np.ascontiguousarray(z_mont, dtype=np.uint64).reshape(4)
        self._chk(self.L.zk_kate_division(self.ctx, p.h, _p(z), (q or p).h), "zk_kate_division")

    def upload_canonical(self, p, data):
        d = _arr(data, 4)
        self._chk(self.L.zk_poly_upload_canonical(self.ctx, p.h, _p(d), d.shape[0]), "zk_poly_upload_canonical")

    # ---- keygen / create_proof -------------------------------------------------------
    def keygen(self, params, fixed_canonical, copies):
        """params: circuit.CircuitParams; fixed_canonical: (n_fix, n, 4) uint64 canonical limbs;
        copies: iterable of ((perm_col, row), (perm_col, row))."""
        cp = CircuitParamsC(params.degree, params.num_advice, params.num_lookup_advice, params.num_fixed, params.lookup_bits,
                            getattr(params, "idle_gate_columns", 0))
        fx = np.ascontiguousarray(fixed_canonical, dtype=np.uint64)
        if fx.ndim != 3 or fx.shape[1:] != (1 << params.degree, 4):
            raise ValueError("fixed_canonical must have shape (n_fixed_columns, 2^degree, 4)")
        cps = np.ascontiguousarray(np.array([[a[0], a[1], b[0], b[1]] for a, b in copies], dtype=np.uint32).reshape(-1, 4))
        h = ctypes.c_uint64()
        self._chk(self.L.zk_keygen(self.ctx, ctypes.byref(cp), _p(fx), fx.shape[0],
                                   cps.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), cps.shape[0], ctypes.byref(h)), "zk_keygen")
        return h.value

    def pk_shape(self, pk):
        out = (ctypes.c_uint32 * 8)()
        self._chk(self.L.zk_pk_shape(self.ctx, pk, out), "zk_pk_shape")
        return dict(zip(("k", "ext_k", "n_advice", "n_fixed", "n_perm", "n_chunks", "n_lookups", "n_h"), out))

    def quotient(self, pk, advice_ext, perm_z_ext, lookup_ext, beta, gamma, y, out_ext, divide=True):
        """Evaluator::evaluate_h (+ divide_by_vanishing_poly) over resident extended cosets; lookup_ext: (a', s', zL) per lookup."""
        arr = lambda ps: (ctypes.c_uint64 * max(len(ps), 1))(*[p.h for p in ps])
        flat = [p for trip in lookup_ext for p in trip]
        f = lambda v: _p(np.ascontiguousarray(v, dtype=np.uint64).reshape(4))
        self._chk(self.L.zk_quotient(self.ctx, pk, arr(advice_ext), len(advice_ext), arr(perm_z_ext), len(perm_z_ext), arr(flat),
                                     len(lookup_ext), f(beta), f(gamma), f(y), 1 if divide else 0, out_ext.h), "zk_quotient")

    def pk_set_transcript_repr(self, pk, repr_mont):
        t = np.ascontiguousarray(repr_mont, dtype=np.uint64).reshape(4)
        self._chk(self.L.zk_pk_set_transcript_repr(self.ctx, pk, _p(t)), "zk_pk_set_transcript_repr")

    def pk_free(self, pk):
        self._chk(self.L.zk_pk_free(self.ctx, pk), "zk_pk_free")

    def vk_export(self, pk):
        counts = (ctypes.c_uint32 * 2)()
        self._chk(self.L.zk_vk_export(self.ctx, pk, None, None, None, counts), "zk_vk_export")
        fc = np.zeros((counts[0], 8), dtype=np.uint64)
        pc = np.zeros((counts[1], 8), dtype=np.uint64)
        tr = np.zeros(4, dtype=np.uint64)
        self._chk(self.L.zk_vk_export(self.ctx, pk, _p(fc), _p(pc), _p(tr), counts), "zk_vk_export")
        return fc, pc, tr

    def prove(self, pk, advice_polys, seed=bytes(32), transcript=ZK_TRANSCRIPT_BLAKE2B, scheme=ZK_SCHEME_DEFAULT):
        if len(seed) != 32:
            raise ValueError("rng seed must be 32 bytes")
        hs = (ctypes.c_uint64 * len(advice_polys))(*[p.h for p in advice_polys])
        ln = ctypes.c_size_t()
        self._chk(self.L.zk_proof_size(self.ctx, pk, transcript, scheme, ctypes.byref(ln)), "zk_proof_size")
        buf = ctypes.create_string_buffer(ln.value)
        self._chk(self.L.zk_prove(self.ctx, pk, hs, len(advice_polys), seed, transcript, scheme, buf, len(buf),
                                  ctypes.byref(ln)), "zk_prove")
        return buf.raw[:ln.value]

    def prove_batch(self, pk, advice_sets, seeds, transcript=ZK_TRANSCRIPT_BLAKE2B, scheme=ZK_SCHEME_DEFAULT):
        """zk_prove_batch: len(advice_sets) independent proofs of one key in lock-step.  advice_sets[j]: proof j's advice
        columns (resident Polys); seeds[j]: its 32-byte RNG seed.  Returns the proofs, each byte-identical to
        prove(pk, advice_sets[j], seeds[j], ...)."""
        B = len(advice_sets)
        if B == 0 or len(seeds) != B or any(len(sd) != 32 for sd in seeds):
            raise ValueError("one 32-byte rng seed per proof")
        na = len(advice_sets[0])
        if any(len(a) != na for a in advice_sets):
            raise ValueError("every proof of a batch has the key's number of advice columns")
        hs = (ctypes.c_uint64 * (B * na))(*[p.h for a in advice_sets for p in a])
        ln = ctypes.c_size_t()
        self._chk(self.L.zk_proof_size(self.ctx, pk, transcript, scheme, ctypes.byref(ln)), "zk_proof_size")
        stride = ln.value
        buf = ctypes.create_string_buffer(stride * B)
        self._chk(self.L.zk_prove_batch(self.ctx, pk, B, hs, na, b"".join(bytes(sd) for sd in seeds), transcript, scheme, buf, stride,
                                        ctypes.byref(ln)), "zk_prove_batch")
        return [buf.raw[j * stride:j * stride + ln.value] for j in range(B)]

    def proof_size(self, pk, transcript=ZK_TRANSCRIPT_BLAKE2B, scheme=ZK_SCHEME_DEFAULT):
        ln = ctypes.c_size_t()
        self._chk(self.L.zk_proof_size(self.ctx, pk, transcript, scheme, ctypes.byref(ln)), "zk_proof_size")
        return ln.value

    def sync(self):
        self._chk(self.L.zk_sync(self.ctx), "zk_sync")

    def timer_reset(self):
        self._chk(self.L.zk_timer_reset(self.ctx), "zk_timer_reset")

    def timer_stats(self, which):
        t, n = ctypes.c_double(), ctypes.c_uint64()
        self._chk(self.L.zk_timer_stats(self.ctx, which, ctypes.byref(t), ctypes.byref(n)), "zk_timer_stats")
        return t.value, n.value

    def audit_report(self):
        """(ordering checks made, violations, description of the first violation) of ZK_OPT_STREAM_AUDIT (zk_audit_report)."""
        counts = (ctypes.c_uint64 * 2)()
        msg = ctypes.create_string_buffer(600)
        self._chk(self.L.zk_audit_report(self.ctx, counts, msg, len(msg)), "zk_audit_report")
        return int(counts[0]), int(counts[1]), msg.value.decode(errors="replace")

    def clock_probe(self, millis=100):
        """(shader-clock ticks, 100 MHz ticks, dependent multiply-adds issued) over ~`millis` ms of one spinning wave (zk_clock_probe)."""
        out = (ctypes.c_uint64 * 4)()
        self._chk(self.L.zk_clock_probe(self.ctx, millis, out), "zk_clock_probe")
        return int(out[0]), int(out[1]), int(out[2])

    def last_ms(self, which):
        v = ctypes.c_float()
        self._chk(self.L.zk_last_kernel_ms(self.ctx, which, ctypes.byref(v)), "zk_last_kernel_ms")
        return v.value
