"""halo2_proofs::transcript::{Blake2bWrite, Blake2bRead, Challenge255} — host side, unchanged by the GPU backend.

Mirrors (halo2_proofs 0.2.0 @ zkwebauthn c254c75, Cargo.lock:1314-1327) src/transcript.rs as the reference uses it at
circuits/src/sgx_dcap_verifier.rs:813 (`Blake2bWrite::<_, _, Challenge255<_>>::init(vec![])`) and :828 (Blake2bRead).
[3P-MEM]: Blake2b-512 personalised "Halo2-Transcript"; absorb prefix bytes 0 (challenge squeeze),
1 (point: x repr then y repr), 2 (scalar repr); `squeeze_challenge` hashes a CLONE of the state after absorbing the
prefix and maps the 64 bytes with `Fr::from_bytes_wide` (little-endian integer mod r).
Points are written compressed: 32-byte little-endian x with the y-parity bit in the top bit of the last byte, identity =
all zero (pasta-style encoding of halo2curves 0.3.1; SURVEY App. B notes the flag position of stack A is unverified —
it does not matter to the GPU library, which only ever returns uncompressed Montgomery coordinates).
This is O(proof size) hashing and stays on the host in the reference as well.
"""
from __future__ import annotations

import hashlib

from .fields import P_MOD, R_MOD

PREFIX_CHALLENGE, PREFIX_POINT, PREFIX_SCALAR = b"\x00", b"\x01", b"\x02"


def point_to_bytes(pt) -> bytes:
    """G1Affine::to_bytes: pt = canonical (x, y) ints or None (identity)."""
    if pt is None:
        return bytes(32)
    x, y = pt
    b = bytearray(x.to_bytes(32, "little"))
    b[31] |= (y & 1) << 7
    return bytes(b)


def point_from_bytes(b: bytes):
    """G1Affine::from_bytes -> (x, y) / None; raises ValueError when x is not on y^2 = x^3 + 3."""
    if len(b) != 32:
        raise ValueError("point encoding must be 32 bytes")
    if b == bytes(32):
        return None
    sign = b[31] >> 7
    x = int.from_bytes(b[:31] + bytes([b[31] & 0x7F]), "little")
    if x >= P_MOD:
        raise ValueError("x coordinate not canonical")
    rhs = (x * x * x + 3) % P_MOD
    y = pow(rhs, (P_MOD + 1) // 4, P_MOD)          # p = 3 mod 4
    if y * y % P_MOD != rhs:
        raise ValueError("not on curve")
    if (y & 1) != sign:
        y = P_MOD - y
    return x, y


class _Blake2bState:
    def __init__(self):
        self.state = hashlib.blake2b(digest_size=64, person=b"Halo2-Transcript")

    def squeeze_challenge(self) -> int:
        """Challenge255: canonical scalar."""
        self.state.update(PREFIX_CHALLENGE)
        return int.from_bytes(self.state.copy().digest(), "little") % R_MOD

    def common_point(self, pt) -> None:
        self.state.update(PREFIX_POINT)
        x, y = (0, 0) if pt is None else pt
        self.state.update(x.to_bytes(32, "little"))
        self.state.update(y.to_bytes(32, "little"))

    def common_scalar(self, s: int) -> None:
        self.state.update(PREFIX_SCALAR)
        self.state.update((s % R_MOD).to_bytes(32, "little"))


class Blake2bWrite(_Blake2bState):
    """`Blake2bWrite::init(vec![])` … `finalize()`."""

    def __init__(self):
        super().__init__()
        self.buf = bytearray()

    def write_point(self, pt) -> None:
        self.common_point(pt)
        self.buf += point_to_bytes(pt)

    def write_scalar(self, s: int) -> None:
        self.common_scalar(s)
        self.buf += (s % R_MOD).to_bytes(32, "little")

    def finalize(self) -> bytes:
        return bytes(self.buf)


class Blake2bRead(_Blake2bState):
    """`Blake2bRead::init(&proof[..])`."""

    def __init__(self, proof: bytes):
        super().__init__()
        self.proof, self.pos = bytes(proof), 0

    def _take(self) -> bytes:
        if self.pos + 32 > len(self.proof):
            raise ValueError("proof too short")
        b = self.proof[self.pos:self.pos + 32]
        self.pos += 32
        return b

    def read_point(self):
        pt = point_from_bytes(self._take())
        self.common_point(pt)
        return pt

    def read_scalar(self) -> int:
        s = int.from_bytes(self._take(), "little")
        if s >= R_MOD:
            raise ValueError("scalar not canonical")
        self.common_scalar(s)
        return s
