"""CPU checks of the oracle's restatement of the reference's file formats (oracle/zkoracle/serde.py)."""
import json
import os

import pytest

from zkoracle import curve as C, field as F, plonk, serde, srs


def test_point_encodings_round_trip():
    pts = [None, C.G1_GEN, C.mul(C.G1_GEN, 5), C.mul(C.G1_GEN, srs.TAU)]
    for fmt in (serde.PROCESSED, serde.RAW_BYTES, serde.RAW_BYTES_UNCHECKED):
        for p in pts:
            assert serde.g1_parse(serde.g1_bytes(p, fmt), fmt) == p
    assert len(serde.g1_bytes(C.G1_GEN, serde.PROCESSED)) == 32 and len(serde.g1_bytes(C.G1_GEN, serde.RAW_BYTES)) == 64
    # (1, 2): y even -> no sign bit; its negative (1, p - 2): y odd -> bit 7 of the last byte
    assert serde.g1_bytes(C.G1_GEN, serde.PROCESSED) == (1).to_bytes(32, "little")
    assert serde.g1_bytes((1, F.P - 2), serde.PROCESSED)[31] == 0x80
    off = next(x for x in range(2, 50) if pow((x ** 3 + 3) % F.P, (F.P - 1) // 2, F.P) != 1)  # no curve point has this x
    with pytest.raises(ValueError):
        serde.g1_parse(off.to_bytes(32, "little"), serde.PROCESSED)
    with pytest.raises(ValueError):
        serde.g1_parse(serde.fq_raw(1) + serde.fq_raw(3), serde.RAW_BYTES)
    assert serde.g1_parse(serde.fq_raw(1) + serde.fq_raw(3), serde.RAW_BYTES_UNCHECKED) == (1, 3)


def test_srs_image_layout_and_known_answers():
    k = 4
    raw = serde.srs_bytes(k, serde.RAW_BYTES)
    n = 1 << k
    assert len(raw) == 4 + 2 * n * 64 + 2 * 128 and raw[:4] == bytes([k, 0, 0, 0])
    assert serde.g1_parse(raw[4:68], serde.RAW_BYTES) == (1, 2)  # g[0] = G1 generator (P256Verifier.yul:777-778)
    assert serde.g1_parse(raw[68:132], serde.RAW_BYTES) == C.mul(C.G1_GEN, srs.TAU)
    lag = srs.lagrange_at(k, srs.TAU)
    assert serde.g1_parse(raw[4 + n * 64:4 + n * 64 + 64], serde.RAW_BYTES) == C.mul(C.G1_GEN, lag[0])
    # K1: the last 128 bytes are [tau]G2 (P256Verifier.yul:1131-1134 stores its negation)
    dec = lambda b: int.from_bytes(b, "little") * F.inv(F.MONT_R, F.P) % F.P
    x0, x1, y0, y1 = (dec(raw[-128 + 32 * i:-96 + 32 * i] if i < 3 else raw[-32:]) for i in range(4))
    assert x1 == 0x0181624E80F3D6AE28DF7E01EAEAB1C0E919877A3B8A6B7FBC69A6817D596EA2
    assert x0 == 0x1783D30DCB12D259BB89098ADDF6280FA4B653BE7A152542A28F7B926E27E648
    assert (-y1) % F.P == 0x00AE44489D41A0D179E2DFDC03BDDD883B7109F8B6AE316A59E815C1A6B35304
    assert (-y0) % F.P == 0x0B2147AB62A386BD63E6DE1522109B8C9588AB466F5AADFDE8C41CA3749423EE
    proc = serde.srs_bytes(k, serde.PROCESSED)
    assert len(proc) == 4 + 2 * n * 32 + 2 * 64
    assert [serde.g1_parse(proc[4 + 32 * i:36 + 32 * i], serde.PROCESSED) for i in range(2 * n)] == \
           [serde.g1_parse(raw[4 + 64 * i:68 + 64 * i], serde.RAW_BYTES) for i in range(2 * n)]


def test_reference_k17_vk_image_round_trips():
    d = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vk_k17.json")))
    pt = lambda q: (int(q[0], 16), int(q[1], 16))
    sh = plonk.Shape(17, 4, 1, 1, 16)
    fc, pc = [pt(q) for q in d["fixed_commitments"]], [pt(q) for q in d["permutation_commitments"]]
    sel = [[(i * (j + 1)) % 3 == 0 for i in range(sh.n)] for j in range(4)]
    for fmt in (serde.PROCESSED, serde.RAW_BYTES):
        img = serde.vk_bytes(sh, fc, pc, sel, fmt)
        assert img[:8] == bytes([0, 0, 0, 17, 0, 0, 0, 6])
        a, b, s = serde.vk_parse(sh, img, fmt)
        assert a == fc and b == pc and s == [[int(v) for v in row] for row in sel]
    with pytest.raises(ValueError):
        serde.vk_parse(plonk.Shape(16, 4, 1, 1, 15), img, serde.RAW_BYTES)
