"""Keccak-f[1600], STROBE-128, Merlin and the Ethereum (Keccak-256) transcript
(oracle; test infrastructure only).

* ``MerlinTranscript`` restates plonk-core/src/transcript.rs:49-109 on top of merlin 3.0
  (third-party, absent from /root/reference; STROBE-128/Keccak-f[1600], protocol label
  "Merlin v1.0").  Pinned by the merlin conformance vector in tests/test_oracle_transcript.py.
* ``EthereumTranscript`` restates gadgets/src/transcript.rs:8-90 and is pinned by the
  reference's own hex KATs (gadgets/src/transcript.rs:101-127).
"""
from __future__ import annotations

from typing import Iterable, Optional, Tuple

from .fields import Curve
from .curve import Point, point_to_bytes_uncompressed

_RC = [
    0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000,
    0x000000000000808B, 0x0000000080000001, 0x8000000080008081, 0x8000000000008009,
    0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
    0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003,
    0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
    0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008,
]
_ROT = [
    [0, 36, 3, 41, 18],
    [1, 44, 10, 45, 2],
    [62, 6, 43, 15, 61],
    [28, 55, 25, 21, 56],
    [27, 20, 39, 8, 14],
]
_M64 = (1 << 64) - 1


def _rol(x: int, n: int) -> int:
    n %= 64
    return ((x << n) | (x >> (64 - n))) & _M64 if n else x


def keccak_f1600(state: bytearray) -> None:
    """In-place Keccak-f[1600] on a 200-byte state (lanes little-endian)."""
    A = [[int.from_bytes(state[8 * (x + 5 * y): 8 * (x + 5 * y) + 8], "little") for y in range(5)]
         for x in range(5)]
    for rnd in range(24):
        C = [A[x][0] ^ A[x][1] ^ A[x][2] ^ A[x][3] ^ A[x][4] for x in range(5)]
        D = [C[(x - 1) % 5] ^ _rol(C[(x + 1) % 5], 1) for x in range(5)]
        A = [[A[x][y] ^ D[x] for y in range(5)] for x in range(5)]
        B = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                B[y][(2 * x + 3 * y) % 5] = _rol(A[x][y], _ROT[x][y])
        A = [[B[x][y] ^ ((~B[(x + 1) % 5][y]) & B[(x + 2) % 5][y]) for y in range(5)]
             for x in range(5)]
        A[0][0] ^= _RC[rnd]
    for x in range(5):
        for y in range(5):
            state[8 * (x + 5 * y): 8 * (x + 5 * y) + 8] = (A[x][y] & _M64).to_bytes(8, "little")


def _sponge(data: bytes, rate: int, pad: int, outlen: int) -> bytes:
    st = bytearray(200)
    msg = bytearray(data)
    msg.append(pad)
    while len(msg) % rate:
        msg.append(0)
    msg[-1] |= 0x80
    for off in range(0, len(msg), rate):
        for i in range(rate):
            st[i] ^= msg[off + i]
        keccak_f1600(st)
    return bytes(st[:outlen])


def keccak256(data: bytes) -> bytes:
    """Legacy Keccak-256 (pad 0x01) as sha3::Keccak256 in gadgets/src/transcript.rs:4."""
    return _sponge(data, 136, 0x01, 32)


def sha3_256(data: bytes) -> bytes:
    """FIPS-202 SHA3-256 (pad 0x06); used only to pin keccak_f1600 against hashlib."""
    return _sponge(data, 136, 0x06, 32)


class Strobe128:
    """STROBE-128 subset used by merlin 3.0 (strobe.rs): AD, meta-AD, PRF, KEY."""
    R = 166
    FLAG_I, FLAG_A, FLAG_C, FLAG_T, FLAG_M, FLAG_K = 1, 2, 4, 8, 16, 32

    def __init__(self, protocol_label: bytes):
        st = bytearray(200)
        st[0:6] = bytes([1, self.R + 2, 1, 0, 1, 96])
        st[6:18] = b"STROBEv1.0.2"
        keccak_f1600(st)
        self.state = st
        self.pos = 0
        self.pos_begin = 0
        self.cur_flags = 0
        self.meta_ad(protocol_label, False)

    def _run_f(self):
        self.state[self.pos] ^= self.pos_begin
        self.state[self.pos + 1] ^= 0x04
        self.state[self.R + 1] ^= 0x80
        keccak_f1600(self.state)
        self.pos = 0
        self.pos_begin = 0

    def _absorb(self, data: bytes):
        for b in data:
            self.state[self.pos] ^= b
            self.pos += 1
            if self.pos == self.R:
                self._run_f()

    def _squeeze(self, n: int) -> bytes:
        out = bytearray()
        for _ in range(n):
            out.append(self.state[self.pos])
            self.state[self.pos] = 0
            self.pos += 1
            if self.pos == self.R:
                self._run_f()
        return bytes(out)

    def _begin_op(self, flags: int, more: bool):
        if more:
            assert self.cur_flags == flags
            return
        assert flags & self.FLAG_T == 0
        old_begin = self.pos_begin
        self.pos_begin = self.pos + 1
        self.cur_flags = flags
        self._absorb(bytes([old_begin, flags]))
        force_f = (flags & (self.FLAG_C | self.FLAG_K)) != 0
        if force_f and self.pos != 0:
            self._run_f()

    def meta_ad(self, data: bytes, more: bool):
        self._begin_op(self.FLAG_M | self.FLAG_A, more)
        self._absorb(data)

    def ad(self, data: bytes, more: bool):
        self._begin_op(self.FLAG_A, more)
        self._absorb(data)

    def prf(self, n: int, more: bool) -> bytes:
        self._begin_op(self.FLAG_I | self.FLAG_A | self.FLAG_C, more)
        return self._squeeze(n)


class Merlin:
    """merlin::Transcript (transcript.rs): new / append_message / append_u64 / challenge_bytes."""

    def __init__(self, label: bytes):
        self.strobe = Strobe128(b"Merlin v1.0")
        self.append_message(b"dom-sep", label)

    def append_message(self, label: bytes, message: bytes):
        self.strobe.meta_ad(label, False)
        self.strobe.meta_ad(len(message).to_bytes(4, "little"), True)
        self.strobe.ad(message, False)

    def append_u64(self, label: bytes, x: int):
        self.append_message(label, int(x).to_bytes(8, "little"))

    def challenge_bytes(self, label: bytes, n: int) -> bytes:
        self.strobe.meta_ad(label, False)
        self.strobe.meta_ad(int(n).to_bytes(4, "little"), True)
        return self.strobe.prf(n, False)


class MerlinTranscript:
    """plonk-core/src/transcript.rs:46-109 (TranscriptProtocol for MerlinTranscript)."""

    def __init__(self, cv: Curve, label: str):
        self.cv = cv
        self.t = Merlin(label.encode())

    def append_u64(self, label: str, item: int):
        self.t.append_u64(label.encode(), item)  # transcript.rs:58-60

    def append_scalar(self, label: str, item: int):
        # transcript.rs:62-67: F::write = canonical integer, little-endian u64 limbs
        self.t.append_message(label.encode(), int(item).to_bytes(self.cv.fr.limbs64 * 8, "little"))

    def append_scalars(self, label: str, items: Iterable[int]):
        # transcript.rs:69-79: one message holding every scalar back to back
        nb = self.cv.fr.limbs64 * 8
        self.t.append_message(label.encode(), b"".join(int(x).to_bytes(nb, "little") for x in items))

    def append_commitment(self, label: str, item: Point):
        # transcript.rs:81-86: PC::Commitment::write = GroupAffine ToBytes (x, y, infinity)
        self.t.append_message(label.encode(), point_to_bytes_uncompressed(self.cv, item))

    def challenge_scalar(self, label: str) -> int:
        # transcript.rs:101-108: (size_in_bits + 7)/8 - 1 bytes -> from_random_bytes (LE integer)
        nbytes = (self.cv.fr.bits + 7) // 8 - 1
        return int.from_bytes(self.t.challenge_bytes(label.encode(), nbytes), "little")


class EthereumTranscript:
    """gadgets/src/transcript.rs:8-90 (BN254 only; labels ignored, big-endian encodings)."""

    def __init__(self, cv: Curve, label: str = ""):
        assert cv.name == "bn254"
        self.cv = cv
        self.state_0 = bytes(32)
        self.state_1 = bytes(32)
        self.counter = 0

    def _append(self, item: bytes):
        old0, old1 = self.state_0, self.state_1
        self.state_0 = keccak256(b"\x00" + old0 + old1 + item)
        self.state_1 = keccak256(b"\x01" + old0 + old1 + item)

    def append_u64(self, label: str, item: int):
        self._append(int(item).to_bytes(8, "big"))

    def append_scalar(self, label: str, item: int):
        self._append(int(item).to_bytes(32, "big"))

    def append_scalars(self, label: str, items: Iterable[int]):
        for it in items:
            self.append_scalar(label, it)

    def append_commitment(self, label: str, item: Point):
        x, y = (0, 1) if item is None else item  # GroupAffine::zero() = (0, 1, inf)
        self._append(int(x).to_bytes(32, "big"))
        self._append(int(y).to_bytes(32, "big"))

    def challenge_scalar(self, label: str) -> int:
        data = b"\x02" + self.state_0 + self.state_1 + self.counter.to_bytes(4, "big")
        self.counter += 1
        q = bytearray(keccak256(data))
        q.reverse()
        q[31] &= 0x1F
        return int.from_bytes(bytes(q), "little")
