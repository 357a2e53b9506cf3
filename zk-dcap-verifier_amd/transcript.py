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


def point_to_bytes(pt, sign_bit: int = 255) -> bytes:
    """G1Affine::to_bytes: pt = canonical (x, y) ints or None (identity).  sign_bit: where the y-parity flag lives — 255 for halo2curves 0.3.1 (stack A,
    identity = all zero), 254 for halo2curves-axiom 0.5.2 (stack B: bit 255 marks the identity there; SURVEY App. B reads this off bin/assets/proof.bin)."""
    if pt is None:
        return bytes(32) if sign_bit == 255 else bytes(31) + b"\x80"
    x, y = pt
    b = bytearray(x.to_bytes(32, "little"))
    b[31] |= (y & 1) << (sign_bit - 248)
    return bytes(b)


def point_from_bytes(b: bytes, sign_bit: int = 255):
    """G1Affine::from_bytes -> (x, y) / None; raises ValueError when x is not on y^2 = x^3 + 3."""
    if len(b) != 32:
        raise ValueError("point encoding must be 32 bytes")
    if sign_bit == 254:
        if b[31] & 0x80:
            if b != bytes(31) + b"\x80":
                raise ValueError("non-canonical identity encoding")
            return None
        sign = (b[31] >> 6) & 1
        x = int.from_bytes(b[:31] + bytes([b[31] & 0x3F]), "little")
        return _lift_x(x, sign)
    if b == bytes(32):
        return None
    sign = b[31] >> 7
    x = int.from_bytes(b[:31] + bytes([b[31] & 0x7F]), "little")
    return _lift_x(x, sign)


def _lift_x(x: int, sign: int):
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
        if pt is None:                                               # Blake2bWrite / Blake2bRead::common_point: `point.coordinates()` is None for the identity
            raise ValueError("cannot write points at infinity to the transcript")
        x, y = pt
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


# ---- stack B: snark-verifier's PoseidonTranscript<G1Affine, NativeLoader, _> (crates/p256-ecdsa/src/base.rs:200-212, bin/src/main.rs:242) --------------------
class _PoseidonState:
    """system/halo2/transcript/halo2.rs ([3P-MEM], see poseidon.py): a point is absorbed as its two coordinates mapped from Fq into Fr (fe_to_fe: the
    integer reduced mod r), a scalar as itself; squeeze_challenge is one sponge squeeze (no domain-separation bytes); points travel compressed with the
    y-parity flag of halo2curves-axiom 0.5.2 (bit 254), scalars as 32-byte little-endian reprs."""
    SIGN_BIT = 254

    def __init__(self):
        from .poseidon import Sponge
        self.sponge = Sponge()

    def squeeze_challenge(self) -> int:
        return self.sponge.squeeze()

    def common_point(self, pt) -> None:
        if pt is None:
            raise ValueError("Invalid elliptic curve point encoding in proof (the identity has no coordinates)")
        self.sponge.update([pt[0] % R_MOD, pt[1] % R_MOD])

    def common_scalar(self, s: int) -> None:
        self.sponge.update([s % R_MOD])


class PoseidonWrite(_PoseidonState):
    """`PoseidonTranscript::<NativeLoader, Vec<u8>>::new::<0>(vec![])` ... `finalize()` — what snark_verifier_sdk::halo2::gen_proof drives (base.rs:200-212)"""

    def __init__(self):
        super().__init__()
        self.buf = bytearray()

    def write_point(self, pt) -> None:
        self.common_point(pt)
        self.buf += point_to_bytes(pt, self.SIGN_BIT)

    def write_scalar(self, s: int) -> None:
        self.common_scalar(s)
        self.buf += (s % R_MOD).to_bytes(32, "little")

    def finalize(self) -> bytes:
        return bytes(self.buf)


class PoseidonRead(_PoseidonState):
    """`PoseidonTranscript::<NativeLoader, &[u8]>::new::<0>(proof)` (bin/src/main.rs:242)"""

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
        pt = point_from_bytes(self._take(), self.SIGN_BIT)
        self.common_point(pt)
        return pt

    def read_scalar(self) -> int:
        s = int.from_bytes(self._take(), "little")
        if s >= R_MOD:
            raise ValueError("scalar not canonical")
        self.common_scalar(s)
        return s


# ---- stack B, EVM flavour: snark-verifier's EvmTranscript<G1Affine, NativeLoader, _, _> (gen_evm_proof_shplonk, crates/p256-ecdsa/src/base.rs:193-199) ------------
class _EvmState:
    """system/halo2/transcript/evm.rs ([3P-MEM]): everything is absorbed as 32-byte BIG-endian words into a byte buffer (a point as x then y, uncompressed);
    squeeze_challenge = keccak256(buffer, plus one byte 0x01 when the buffer is exactly the 32 bytes of the previous hash), the hash becomes the new buffer
    and, read as a big-endian integer reduced mod r, the challenge.  Points travel as 64 bytes, scalars as 32, all big endian — the layout the generated
    Solidity verifier reads from calldata."""

    def __init__(self):
        self.buf = bytearray()

    def squeeze_challenge(self) -> int:
        from .keccak import keccak256
        data = bytes(self.buf) + (b"\x01" if len(self.buf) == 32 else b"")
        h = keccak256(data)
        self.buf = bytearray(h)
        return int.from_bytes(h, "big") % R_MOD

    def common_point(self, pt) -> None:
        if pt is None:
            raise ValueError("Invalid elliptic curve point encoding in proof (the identity has no coordinates)")
        self.buf += pt[0].to_bytes(32, "big") + pt[1].to_bytes(32, "big")

    def common_scalar(self, s: int) -> None:
        self.buf += (s % R_MOD).to_bytes(32, "big")


class EvmWrite(_EvmState):
    def __init__(self):
        super().__init__()
        self.out = bytearray()

    def write_point(self, pt) -> None:
        self.common_point(pt)
        self.out += pt[0].to_bytes(32, "big") + pt[1].to_bytes(32, "big")

    def write_scalar(self, s: int) -> None:
        self.common_scalar(s)
        self.out += (s % R_MOD).to_bytes(32, "big")

    def finalize(self) -> bytes:
        return bytes(self.out)


class EvmRead(_EvmState):
    def __init__(self, proof: bytes):
        super().__init__()
        self.proof, self.pos = bytes(proof), 0

    def _take(self, n: int) -> bytes:
        if self.pos + n > len(self.proof):
            raise ValueError("proof too short")
        self.pos += n
        return self.proof[self.pos - n:self.pos]

    def read_point(self):
        b = self._take(64)
        x, y = int.from_bytes(b[:32], "big"), int.from_bytes(b[32:], "big")
        if x >= P_MOD or y >= P_MOD or (y * y - x * x * x - 3) % P_MOD:
            raise ValueError("not on curve")
        self.common_point((x, y))
        return x, y

    def read_scalar(self) -> int:
        s = int.from_bytes(self._take(32), "big")
        if s >= R_MOD:
            raise ValueError("scalar not canonical")
        self.common_scalar(s)
        return s
