"""Keccak-256 (original padding, as the EVM's KECCAK256) and ChaCha20 keystream.

Oracle (test infrastructure).  Keccak is what snark-verifier's EvmTranscript
hashes with (pinned by reference proving-server/P256Verifier.yul:75,97,104);
ChaCha20 is rand_chacha's ChaCha20Rng, which `gen_srs` seeds with [0;32]
(call sites reference halo2-circuits/src/ecc/ecdsa_p256.rs:258,338).
"""
import struct

_M64 = (1 << 64) - 1
_RC = [
    0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000,
    0x000000000000808B, 0x0000000080000001, 0x8000000080008081, 0x8000000000008009,
    0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
    0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003,
    0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
    0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008,
]
_ROT = [
    [0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61],
    [28, 55, 25, 21, 56], [27, 20, 39, 8, 14],
]


def _rol(x, n):
    n %= 64
    return ((x << n) | (x >> (64 - n))) & _M64 if n else x


def _keccak_f(A):
    for rc in _RC:
        C = [A[x][0] ^ A[x][1] ^ A[x][2] ^ A[x][3] ^ A[x][4] for x in range(5)]
        D = [C[(x - 1) % 5] ^ _rol(C[(x + 1) % 5], 1) for x in range(5)]
        A = [[A[x][y] ^ D[x] for y in range(5)] for x in range(5)]
        Bm = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                Bm[y][(2 * x + 3 * y) % 5] = _rol(A[x][y], _ROT[x][y])
        A = [[Bm[x][y] ^ ((~Bm[(x + 1) % 5][y]) & Bm[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
        A[0][0] ^= rc
    return A


def keccak256(data: bytes) -> bytes:
    rate = 136
    msg = bytearray(data)
    msg.append(0x01)
    while len(msg) % rate:
        msg.append(0)
    msg[-1] |= 0x80
    A = [[0] * 5 for _ in range(5)]
    for off in range(0, len(msg), rate):
        blk = msg[off:off + rate]
        for i in range(rate // 8):
            A[i % 5][i // 5] ^= struct.unpack_from("<Q", blk, 8 * i)[0]
        A = _keccak_f(A)
    out = b"".join(struct.pack("<Q", A[i % 5][i // 5]) for i in range(4))
    return out


# ---- ChaCha20 ---------------------------------------------------------------

def _qr(s, a, b, c, d):
    M = 0xFFFFFFFF
    s[a] = (s[a] + s[b]) & M; s[d] ^= s[a]; s[d] = ((s[d] << 16) | (s[d] >> 16)) & M
    s[c] = (s[c] + s[d]) & M; s[b] ^= s[c]; s[b] = ((s[b] << 12) | (s[b] >> 20)) & M
    s[a] = (s[a] + s[b]) & M; s[d] ^= s[a]; s[d] = ((s[d] << 8) | (s[d] >> 24)) & M
    s[c] = (s[c] + s[d]) & M; s[b] ^= s[c]; s[b] = ((s[b] << 7) | (s[b] >> 25)) & M


def chacha20_block(key: bytes, counter: int, stream: int = 0) -> bytes:
    """One 64-byte block; words 12-13 = 64-bit block counter, 14-15 = stream id
    (rand_chacha layout)."""
    st = list(struct.unpack("<4I", b"expand 32-byte k")) + list(struct.unpack("<8I", key))
    st += [counter & 0xFFFFFFFF, (counter >> 32) & 0xFFFFFFFF, stream & 0xFFFFFFFF, (stream >> 32) & 0xFFFFFFFF]
    w = st[:]
    for _ in range(10):
        _qr(w, 0, 4, 8, 12); _qr(w, 1, 5, 9, 13); _qr(w, 2, 6, 10, 14); _qr(w, 3, 7, 11, 15)
        _qr(w, 0, 5, 10, 15); _qr(w, 1, 6, 11, 12); _qr(w, 2, 7, 8, 13); _qr(w, 3, 4, 9, 14)
    return struct.pack("<16I", *[(w[i] + st[i]) & 0xFFFFFFFF for i in range(16)])


class ChaCha20Rng:
    """Byte stream of ChaCha20(key, nonce 0) from block 0; `fr()` restates
    halo2curves `Fr::random` = from_u512(8 x next_u64)."""

    def __init__(self, seed: bytes):
        assert len(seed) == 32
        self.key = seed
        self.block = 0
        self.buf = b""

    def bytes(self, n):
        while len(self.buf) < n:
            self.buf += chacha20_block(self.key, self.block)
            self.block += 1
        out, self.buf = self.buf[:n], self.buf[n:]
        return out

    def fr(self):
        from .field import R
        return int.from_bytes(self.bytes(64), "little") % R
