"""The reference's on-disk artefacts restated byte by byte: SRS (`ParamsKZG::write`), verifying key
(`VerifyingKey::write`) and proving key (`ProvingKey::write`) in halo2_proofs' three `SerdeFormat`s.

Oracle (test infrastructure) — see oracle/zkoracle/__init__.py.  Written by the reference at
halo2-circuits/src/ecc/ecdsa_p256.rs:261-270 (keys, SerdeFormat::RawBytes) and by halo2-base `gen_srs`
(./params/kzg_bn254_{k}.srs), read back on every request at :338-343, :388-393.  The layouts live in halo2_proofs
(PSE fork; NOT under /root/reference): poly/kzg/commitment.rs `write_custom`, plonk.rs `VerifyingKey::write` /
`ProvingKey::write`, plonk/permutation.rs, poly.rs `Polynomial::write`, helpers.rs — restated [RECALLED]:

  Processed          field elements: canonical 32-byte little-endian; G1: x little-endian, bit 7 of the last byte =
                     y odd, identity = 32 zero bytes; G2: x.c0 || x.c1, sign of y.c0 in bit 7 of the last byte
  RawBytes(Unchecked) the Rust memory image: Montgomery limbs little-endian; G1Affine x || y; G2Affine x.c0 || x.c1 || y.c0 || y.c1
  ParamsKZG          u32 LE k | g | g_lagrange | g2 | s_g2
  Polynomial         u32 BE len | values;  slice: u32 BE count | polynomials
  VerifyingKey       u32 BE k | u32 BE #fixed | fixed commitments | permutation commitments | selectors (2^k bits
                     each, packed LSB-first)
  ProvingKey         vk | l0 | l_last | l_active_row | fixed_values | fixed_polys | fixed_cosets | permutations | polys | cosets

Parity status: no bytes of these files exist anywhere in the reference, so the layouts are unpinned beyond the known
answers they carry — s_g2 of the SRS (P256Verifier.yul:1131-1134) and the twelve k=17 vk commitments
(yul:880-980); the product (csrc/serde.hip) and this restatement come from the same recollection.
"""
import numpy as np

from . import cops, curve as C
from .field import MONT_R, P, R, inv, omega
from .srs import TAU

PROCESSED, RAW_BYTES, RAW_BYTES_UNCHECKED = 0, 1, 2


def _le(v):
    return int(v).to_bytes(32, "little")


def fq_raw(v):
    return _le(v * MONT_R % P)


def g1_bytes(pt, fmt):
    """pt: (x, y) canonical ints or None (identity)."""
    if fmt == PROCESSED:
        if pt is None:
            return bytes(32)
        b = bytearray(_le(pt[0]))
        b[31] |= (pt[1] & 1) << 7
        return bytes(b)
    if pt is None:
        return bytes(64)
    return fq_raw(pt[0]) + fq_raw(pt[1])


def g1_parse(b, fmt):
    if fmt == PROCESSED:
        x = int.from_bytes(b, "little")
        sign, x = x >> 255, x & ((1 << 255) - 1)
        if x >= P:
            raise ValueError("non-canonical x")
        if x == 0 and not sign:
            return None
        rhs = (x * x * x + 3) % P
        y = pow(rhs, (P + 1) // 4, P)
        if y * y % P != rhs:
            raise ValueError("not on the curve")
        return (x, y if (y & 1) == sign else P - y)
    xm, ym = int.from_bytes(b[:32], "little"), int.from_bytes(b[32:], "little")
    if fmt == RAW_BYTES and (xm >= P or ym >= P):
        raise ValueError("non-canonical coordinate")
    x, y = xm * inv(MONT_R, P) % P, ym * inv(MONT_R, P) % P
    if x == 0 and y == 0:
        return None
    if fmt == RAW_BYTES and (y * y - x * x * x - 3) % P:
        raise ValueError("not on the curve")
    return (x, y)


def g2_bytes(pt, fmt):
    """pt: ((x.c0, x.c1), (y.c0, y.c1)) canonical ints."""
    (x0, x1), (y0, y1) = pt
    if fmt == PROCESSED:
        b = bytearray(_le(x0) + _le(x1))
        b[63] |= (y0 & 1) << 7
        return bytes(b)
    return fq_raw(x0) + fq_raw(x1) + fq_raw(y0) + fq_raw(y1)


def fr_vec_bytes(a_mont, fmt):
    """(n, 4) Montgomery array -> n * 32 bytes."""
    a = np.ascontiguousarray(a_mont, dtype=np.uint64)
    return (cops.from_mont_arr(a) if fmt == PROCESSED else a).tobytes()


def affine_arr_bytes(pts_mont, fmt):
    """(n, 8) Montgomery affine array -> bytes."""
    if fmt != PROCESSED:
        return np.ascontiguousarray(pts_mont, dtype=np.uint64).tobytes()
    return b"".join(g1_bytes(pt, fmt) for pt in cops.affine_arr_to_ints(pts_mont))


# ------------------------------------------------------------------- SRS ---

def srs_arrays(k):
    """g, g_lagrange of gen_srs(k) as (n, 8) Montgomery affine arrays (ParamsKZG::setup with the secret known)."""
    n = 1 << k
    g = cops.fixed_base_g1(cops.fr_powers(TAU, n))
    w = cops.fr_powers(omega(k), n)
    c = (pow(TAU, n, R) - 1) * inv(n, R) % R
    lag = [wi * c % R * inv((TAU - wi) % R, R) % R for wi in cops.fr_ints(w)]
    return g, cops.fixed_base_g1(cops.fr_mont(lag))


def srs_bytes(k, fmt, g=None, gl=None):
    if g is None:
        g, gl = srs_arrays(k)
    s_g2 = C.g2_mul(C.G2_GEN, TAU)
    return int(k).to_bytes(4, "little") + affine_arr_bytes(g, fmt) + affine_arr_bytes(gl, fmt) + g2_bytes(C.G2_GEN, fmt) + g2_bytes(s_g2, fmt)


# --------------------------------------------------------------- vk / pk ---

def be32(v):
    return int(v).to_bytes(4, "big")


def selectors_of(shape, fixed):
    """Selector vectors of the halo2-lib shape in creation order: q_enable per gate column, then (single-column
    strategy) q_lookup; `fixed`: [n_fix][n] ints.  A never-enabled gate column's selector is all-false."""
    n = shape.n
    sel = [[0] * n if f is None else [1 if v else 0 for v in fixed[f]] for f in shape.fx_sel]
    if shape.single:
        sel.append([1 if v else 0 for v in fixed[shape.fx_qlookup]])
    return sel


def pack_bits(bits):
    out = bytearray(len(bits) // 8)
    for i, b in enumerate(bits):
        if b:
            out[i >> 3] |= 1 << (i & 7)
    return bytes(out)


def vk_bytes(shape, fixed_commitments, permutation_commitments, selectors, fmt):
    """`fixed_commitments` in the oracle's (query) order; the file lists them in halo2's column order (table first)."""
    from .vkrepr import halo2_fixed_order

    out = be32(shape.k) + be32(len(fixed_commitments))
    out += b"".join(g1_bytes(fixed_commitments[i], fmt) for i in halo2_fixed_order(shape))
    out += b"".join(g1_bytes(p, fmt) for p in permutation_commitments)
    out += b"".join(pack_bits(s) for s in selectors)
    return out


def vk_parse(shape, b, fmt):
    """-> (fixed commitments, permutation commitments, selector bit vectors); raises on a malformed image."""
    gs = 32 if fmt == PROCESSED else 64
    n_sel = shape.n_gate + (1 if shape.single else 0)
    if len(b) != 8 + (shape.n_fix + len(shape.perm_cols)) * gs + n_sel * (shape.n // 8):
        raise ValueError("bad vk length")
    if int.from_bytes(b[:4], "big") != shape.k or int.from_bytes(b[4:8], "big") != shape.n_fix:
        raise ValueError("vk header does not match the shape")
    pos = 8
    pts = []
    for _ in range(shape.n_fix + len(shape.perm_cols)):
        pts.append(g1_parse(b[pos:pos + gs], fmt))
        pos += gs
    sels = []
    for _ in range(n_sel):
        raw = b[pos:pos + shape.n // 8]
        sels.append([(raw[i >> 3] >> (i & 7)) & 1 for i in range(shape.n)])
        pos += shape.n // 8
    from .vkrepr import halo2_fixed_order

    fixed = [None] * shape.n_fix
    for pos, i in enumerate(halo2_fixed_order(shape)):
        fixed[i] = pts[pos]
    return fixed, pts[shape.n_fix:], sels


def poly_bytes(a_mont, fmt):
    return be32(a_mont.shape[0]) + fr_vec_bytes(a_mont, fmt)


def slice_bytes(polys, fmt):
    return be32(len(polys)) + b"".join(poly_bytes(p, fmt) for p in polys)


def pk_bytes(fpk, fixed_ints, fmt):
    """ProvingKey::write of a zkoracle.fastprover key (`fixed_ints`: the fixed columns as ints, for the selectors)."""
    from . import fastprover as fp

    sh = fpk.shape
    one = np.tile(fp.m1(1), (1 << sh.ext_k, 1))
    l_active = fp.lin(fp.lin(one, 1, fpk.llast_e, R - 1), 1, fpk.lblind_e, R - 1)  # 1 - l_last - l_blind
    out = vk_bytes(sh, fpk.vk.fixed_commitments, fpk.vk.permutation_commitments, selectors_of(sh, fixed_ints), fmt)
    out += poly_bytes(fpk.l0_e, fmt) + poly_bytes(fpk.llast_e, fmt) + poly_bytes(l_active, fmt)
    from .vkrepr import halo2_fixed_order

    order = halo2_fixed_order(sh)  # fixed columns in halo2's column order (table first)
    out += slice_bytes([fpk.fixed[i] for i in order], fmt) + slice_bytes([fpk.fix_c[i] for i in order], fmt) + slice_bytes([fpk.fix_e[i] for i in order], fmt)
    out += slice_bytes(fpk.sigma, fmt) + slice_bytes(fpk.sig_c, fmt) + slice_bytes(fpk.sig_e, fmt)
    return out
