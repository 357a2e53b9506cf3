"""TEST INFRASTRUCTURE ONLY — second writing of Keccak-256 and of snark-verifier's EvmTranscript reader (checker for zk-dcap-verifier_amd/{keccak,transcript}.py).
Keccak-f[1600] is pinned by the public known answers (tests/test_evm_transcript.py); the transcript framing is [3P-MEM] (crates pinned at Cargo.lock:2577-2618,
reached from crates/p256-ecdsa/src/base.rs:193-199): agreement of the two writings does not pin it to the Rust crate."""
import pyref as p

_M = 0xFFFFFFFFFFFFFFFF


def _rounds(lanes):
    """flat 25-lane state, index x + 5 y; round constants from the LFSR of the specification instead of a table"""
    r = 1
    for _ in range(24):
        c = [lanes[x] ^ lanes[x + 5] ^ lanes[x + 10] ^ lanes[x + 15] ^ lanes[x + 20] for x in range(5)]
        for x in range(5):
            t = c[(x + 4) % 5] ^ (((c[(x + 1) % 5] << 1) | (c[(x + 1) % 5] >> 63)) & _M)
            for y in range(0, 25, 5):
                lanes[x + y] ^= t
        x, y, cur = 1, 0, lanes[1]
        for t in range(24):                                          # rho and pi along the (x, y) -> (y, 2x + 3y) orbit
            x, y = y, (2 * x + 3 * y) % 5
            sh = ((t + 1) * (t + 2) // 2) % 64
            cur, lanes[x + 5 * y] = lanes[x + 5 * y], ((cur << sh) | (cur >> (64 - sh))) & _M
        for y in range(0, 25, 5):
            row = lanes[y:y + 5]
            for x in range(5):
                lanes[y + x] = row[x] ^ ((~row[(x + 1) % 5]) & row[(x + 2) % 5] & _M)
        for j in range(7):                                           # iota
            r = ((r << 1) ^ ((r >> 7) * 0x71)) % 256
            if r & 2:
                lanes[0] ^= 1 << ((1 << j) - 1)
    return lanes


def keccak256(data: bytes) -> bytes:
    m = bytearray(data) + b"\x01"
    m += bytes(-len(m) % 136)
    m[-1] ^= 0x80
    lanes = [0] * 25
    for off in range(0, len(m), 136):
        for i in range(17):
            lanes[i] ^= int.from_bytes(m[off + 8 * i:off + 8 * i + 8], "little")
        lanes = _rounds(lanes)
    return b"".join(v.to_bytes(8, "little") for v in lanes[:4])


class Reader:
    """transcript reader interface of oracle/verifier.py over the EVM transcript"""

    def __init__(self, proof: bytes):
        self.acc = b""
        self.proof, self.pos = bytes(proof), 0

    def squeeze(self) -> int:
        h = keccak256(self.acc + (b"\x01" if len(self.acc) == 32 else b""))
        self.acc = h
        return int.from_bytes(h, "big") % p.R

    def common_scalar(self, s: int):
        self.acc += (s % p.R).to_bytes(32, "big")

    def common_point(self, pt):
        self.acc += pt[0].to_bytes(32, "big") + pt[1].to_bytes(32, "big")

    def _take(self, n):
        if self.pos + n > len(self.proof):
            raise ValueError("proof too short")
        self.pos += n
        return self.proof[self.pos - n:self.pos]

    def read_point(self):
        b = self._take(64)
        pt = (int.from_bytes(b[:32], "big"), int.from_bytes(b[32:], "big"))
        if pt[0] >= p.P or pt[1] >= p.P or not p.g1_is_on_curve(pt):
            raise ValueError("commitment not on the curve")
        self.common_point(pt)
        return pt

    def read_scalar(self) -> int:
        s = int.from_bytes(self._take(32), "big")
        if s >= p.R:
            raise ValueError("evaluation not canonical")
        self.common_scalar(s)
        return s
