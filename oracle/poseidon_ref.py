"""TEST INFRASTRUCTURE ONLY — second, independently written Poseidon sponge + transcript reader for the stack-B (PoseidonTranscript) proofs.

Checker for zk-dcap-verifier_amd/{poseidon,transcript}.py: same published algorithm (Poseidon permutation over bn256::Fr with x^5, T = 3, RATE = 2,
R_F = 8, R_P = 57; Grain-LFSR constants and Cauchy MDS as the `poseidon` crate derives them; snark-verifier's sponge and PoseidonTranscript — crates
pinned at Cargo.lock:2577-2618, reached from crates/p256-ecdsa/src/base.rs:200-212 and bin/src/main.rs:242), written separately so that a slip of the
pen in one of them shows up as a mismatch.  Both come from memory of crates that are not on this machine ([3P-MEM]): agreement of the two does NOT pin
either to the Rust implementation — parity with it is unpinned.
"""
import itertools

import pyref as p

R = p.R


def _grain_bits(t, r_f, r_p, n_bits=254):
    init = "{:02b}{:04b}{:012b}{:012b}{:010b}{:010b}".format(1, 0, n_bits, t, r_f, r_p) + "1" * 30
    s = [int(c) for c in init]

    def clock():
        b = s[0] ^ s[13] ^ s[23] ^ s[38] ^ s[51] ^ s[62]
        del s[0]
        s.append(b)
        return b
    for _ in range(160):
        clock()
    while True:
        keep, bit = clock(), clock()
        if keep:
            yield bit


def constants(t=3, r_f=8, r_p=57, skip_mds=0):
    bits = _grain_bits(t, r_f, r_p)
    take = lambda: int("".join(str(b) for b in itertools.islice(bits, 254)), 2)
    rc = []
    while len(rc) < (r_f + r_p) * t:
        v = take()
        if v < R:
            rc.append(v)
    while True:
        vals = [take() % R for _ in range(2 * t)]
        if len(set(vals)) != 2 * t:
            continue
        if skip_mds:
            skip_mds -= 1
            continue
        m = [[pow(vals[i] + vals[t + j], -1, R) for j in range(t)] for i in range(t)]
        return [rc[i * t:(i + 1) * t] for i in range(r_f + r_p)], m


_C = constants()


def permutation(st):
    rc, m = _C
    st = list(st)
    for r, row in enumerate(rc):
        st = [(a + b) % R for a, b in zip(st, row)]
        full = r < 4 or r >= 4 + 57
        st = [pow(v, 5, R) if (full or i == 0) else v for i, v in enumerate(st)]
        st = [sum(a * b for a, b in zip(mrow, st)) % R for mrow in m]
    return st


class Reader:
    """the interface oracle/verifier.py's verify_proof expects of its transcript reader, over the Poseidon sponge"""

    def __init__(self, proof: bytes):
        self.st = [1 << 64, 0, 0]
        self.pending = []
        self.proof, self.pos = bytes(proof), 0

    def squeeze(self) -> int:
        q, self.pending = self.pending, []
        chunks = [q[i:i + 2] for i in range(0, len(q), 2)]
        if len(q) % 2 == 0:
            chunks.append([])
        for ch in chunks:
            for i, v in enumerate(ch):
                self.st[i + 1] = (self.st[i + 1] + v) % R
            if len(ch) < 2:
                self.st[len(ch) + 1] = (self.st[len(ch) + 1] + 1) % R
            self.st = permutation(self.st)
        return self.st[1]

    def common_scalar(self, s: int):
        self.pending.append(s % R)

    def common_point(self, pt):
        if pt is None:
            raise ValueError("identity has no coordinates")
        self.pending += [pt[0] % R, pt[1] % R]

    def _take(self) -> bytes:
        if self.pos + 32 > len(self.proof):
            raise ValueError("proof too short")
        self.pos += 32
        return self.proof[self.pos - 32:self.pos]

    def read_point(self):
        b = self._take()
        if b[31] & 0x80:
            raise ValueError("identity / non-canonical point in proof")
        x = int.from_bytes(b, "little") & ((1 << 254) - 1)
        ys = p.g1_decompress_x(x) if x < p.P else None
        if ys is None:
            raise ValueError("commitment not on the curve")
        y = ys[0] if (ys[0] & 1) == ((b[31] >> 6) & 1) else ys[1]
        self.common_point((x, y))
        return (x, y)

    def read_scalar(self) -> int:
        s = int.from_bytes(self._take(), "little")
        if s >= R:
            raise ValueError("evaluation not canonical")
        self.common_scalar(s)
        return s
