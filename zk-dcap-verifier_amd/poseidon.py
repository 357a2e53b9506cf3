"""Poseidon over bn256::Fr as stack B's transcript hash uses it — host side, O(proof size) work.

The reference's second proving stack (crates/p256-ecdsa/src/base.rs:193-212, bin/src/main.rs:233-253) writes and reads its proofs through
`snark_verifier_sdk::halo2::PoseidonTranscript<NativeLoader, _>` (`PoseidonTranscript::<NativeLoader, &[u8]>::new::<0>(proof)` at main.rs:242):
snark-verifier 0.1.7 @ axiom-crypto 4b733e0 (Cargo.lock:2577-2618), `snark-verifier/src/util/hash/poseidon.rs` + `system/halo2/transcript/halo2.rs`,
with the SDK's parameters T = 3, RATE = 2, R_F = 8, R_P = 57, SECURE_MDS = 0 (`snark-verifier-sdk/src/lib.rs`).

[3P-MEM] — restated from the published algorithm (Grassi et al., Poseidon; round constants and the Cauchy MDS matrix from the Grain LFSR exactly as
the `poseidon` crate of privacy-scaling-explorations generates them: grain.rs / mds.rs).  The crate is not on this machine and no vector of the
reference pins it: the reference's one Poseidon artefact, bin/assets/proof.bin, needs params/vk.bin to be replayed, which is git-ignored.
Two independent writings of the same recollection (this file and the test suite's own checker) are cross-checked in tests/test_poseidon_transcript.py;
equality with the Rust crate stays UNPINNED until one `cargo test` dumps a squeeze (shim/README.md).
"""
from __future__ import annotations

from typing import List

from .fields import R_MOD

T, RATE, R_F, R_P, SECURE_MDS = 3, 2, 8, 57, 0
NUM_BITS = 254


class Grain:
    """The Poseidon paper's Grain LFSR in self-shrinking mode (poseidon crate, grain.rs): 80-bit state initialised with
    field type (2 bits, 1 = prime), s-box (4 bits, 0 = x^alpha), field size (12), t (12), R_F (10), R_P (10), thirty 1 bits; 160 bits discarded."""

    def __init__(self, t: int, r_f: int, r_p: int, field_bits: int = NUM_BITS):
        bits: List[int] = []
        for value, length in ((1, 2), (0, 4), (field_bits, 12), (t, 12), (r_f, 10), (r_p, 10)):
            bits += [(value >> (length - 1 - i)) & 1 for i in range(length)]         # MSB first
        bits += [1] * 30
        assert len(bits) == 80
        self.state = bits
        for _ in range(160):
            self._raw()

    def _raw(self) -> int:
        s = self.state
        new = s[62] ^ s[51] ^ s[38] ^ s[23] ^ s[13] ^ s[0]
        s.pop(0)
        s.append(new)
        return new

    def bit(self) -> int:
        while True:                                                    # self-shrinking: a pair (1, b) emits b, a pair (0, b) emits nothing
            b1, b2 = self._raw(), self._raw()
            if b1:
                return b2

    def _integer(self) -> int:
        v = 0
        for _ in range(NUM_BITS):                                      # the reference implementation reads the bits MSB first
            v = (v << 1) | self.bit()
        return v

    def next_field_element(self) -> int:
        while True:                                                    # round constants: rejection sampling
            v = self._integer()
            if v < R_MOD:
                return v

    def next_field_element_without_rejection(self) -> int:
        return self._integer() % R_MOD                                # MDS sampling: from_bytes_wide


def generate(t: int = T, r_f: int = R_F, r_p: int = R_P, secure_mds: int = SECURE_MDS):
    """-> (round_constants [(r_f + r_p)][t], mds [t][t])"""
    g = Grain(t, r_f, r_p)
    rc = [[g.next_field_element() for _ in range(t)] for _ in range(r_f + r_p)]
    select = secure_mds
    while True:
        while True:
            vals = [g.next_field_element_without_rejection() for _ in range(2 * t)]
            if len(set(vals)) == len(vals):
                break
        if select:
            select -= 1
            continue
        xs, ys = vals[:t], vals[t:]
        mds = [[pow((x + y) % R_MOD, -1, R_MOD) for y in ys] for x in xs]
        return rc, mds


_SPEC = None


def spec():
    global _SPEC
    if _SPEC is None:
        _SPEC = generate()
    return _SPEC


def permute(state: List[int]) -> List[int]:
    """the Poseidon permutation (x^5 s-box): R_F / 2 full rounds, R_P partial rounds (s-box on word 0), R_F / 2 full rounds; each round adds its
    constants, applies the s-box, multiplies by the MDS matrix.  (snark-verifier runs the 'optimized' schedule with sparse matrices: same function.)"""
    rc, mds = spec()
    t = len(state)
    half = R_F // 2
    for r in range(R_F + R_P):
        state = [(s + c) % R_MOD for s, c in zip(state, rc[r])]
        if r < half or r >= half + R_P:
            state = [pow(s, 5, R_MOD) for s in state]
        else:
            state[0] = pow(state[0], 5, R_MOD)
        state = [sum(mds[i][j] * state[j] for j in range(t)) % R_MOD for i in range(t)]
    return state


class Sponge:
    """snark-verifier `Poseidon<F, L, T, RATE>`: update() buffers, squeeze() absorbs RATE elements per permutation (a short chunk is padded with one 1;
    a buffer that is a whole number of chunks — the empty buffer included — gets one more permutation of an empty chunk) and returns state[1]."""

    def __init__(self):
        self.state = [1 << 64] + [0] * (T - 1)
        self.buf: List[int] = []

    def update(self, elements) -> None:
        self.buf.extend(int(e) % R_MOD for e in elements)

    def _absorb(self, chunk) -> None:
        for i, v in enumerate(chunk):
            self.state[1 + i] = (self.state[1 + i] + v) % R_MOD
        if len(chunk) < RATE:
            self.state[1 + len(chunk)] = (self.state[1 + len(chunk)] + 1) % R_MOD
        self.state = permute(self.state)

    def squeeze(self) -> int:
        buf, self.buf = self.buf, []
        for i in range(0, len(buf), RATE):
            self._absorb(buf[i:i + RATE])
        if len(buf) % RATE == 0:
            self._absorb([])
        return self.state[1]
