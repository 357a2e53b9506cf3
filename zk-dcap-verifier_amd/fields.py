"""Host-side BN254 scalar / base field helpers for the prover's O(proof size) logic (challenges, evaluation
points, point encoding) — the part of halo2's `create_proof` that stays on the host in the reference too
(SURVEY.md §8a rows a16/a17).  Bulk arithmetic never goes through here: columns live in HBM as raw Montgomery
limbs (halo2curves::bn256::{Fr, Fq} layout, [u64; 4] little endian, R = 2^256).
"""
from __future__ import annotations

import numpy as np

R_MOD = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001   # Fr
P_MOD = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47   # Fq
S = 28
MULT_GEN = 7
ROOT_OF_UNITY = pow(MULT_GEN, (R_MOD - 1) >> S, R_MOD)
DELTA = pow(MULT_GEN, 1 << S, R_MOD)
ZETA = pow(MULT_GEN, (R_MOD - 1) // 3, R_MOD)
_MASK = 0xFFFFFFFFFFFFFFFF
_RINV_R = pow(1 << 256, -1, R_MOD)
_RINV_P = pow(1 << 256, -1, P_MOD)


def limbs(v: int) -> np.ndarray:
    return np.array([(v >> (64 * i)) & _MASK for i in range(4)], dtype=np.uint64)


def unlimbs(a) -> int:
    a = np.asarray(a, dtype=np.uint64).reshape(-1)
    return sum(int(a[i]) << (64 * i) for i in range(4))


def fr_mont(x: int) -> np.ndarray:
    """canonical int -> Montgomery limbs (what Rust holds for an Fr)."""
    return limbs(((x % R_MOD) << 256) % R_MOD)


def fr_int(a) -> int:
    """Montgomery limbs -> canonical int."""
    return unlimbs(a) * _RINV_R % R_MOD


def fq_mont(x: int) -> np.ndarray:
    return limbs(((x % P_MOD) << 256) % P_MOD)


def fq_int(a) -> int:
    return unlimbs(a) * _RINV_P % P_MOD


def fr_mont_array(vals) -> np.ndarray:
    """list of canonical ints -> (n, 4) uint64 Montgomery array."""
    out = np.empty((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        m = ((int(v) % R_MOD) << 256) % R_MOD
        out[i, 0] = m & _MASK
        out[i, 1] = (m >> 64) & _MASK
        out[i, 2] = (m >> 128) & _MASK
        out[i, 3] = m >> 192
    return out


def fr_int_array(a) -> list:
    b = np.ascontiguousarray(a, dtype="<u8").tobytes()
    return [int.from_bytes(b[i:i + 32], "little") * _RINV_R % R_MOD for i in range(0, len(b), 32)]


def omega(k: int) -> int:
    return pow(ROOT_OF_UNITY, 1 << (S - k), R_MOD)


def g1_affine_ints(jac12) -> tuple | None:
    """normalised G1 {x, y, z} (12 limbs; z = mont(1) or all zero) from zk_msm -> canonical affine (x, y) or None."""
    j = np.asarray(jac12, dtype=np.uint64).reshape(12)
    if not j[8:12].any():
        return None
    return fq_int(j[0:4]), fq_int(j[4:8])


_R_L = [np.uint64((R_MOD >> (64 * i)) & _MASK) for i in range(4)]


def _below_r(a: np.ndarray) -> np.ndarray:
    """row-wise (n, 4) little-endian limbs < r"""
    top = a[:, 3]
    lt = top < _R_L[3]
    eq = np.nonzero(top == _R_L[3])[0]                 # 2^-62 of the draws: decided by the lower limbs
    for j in eq:
        lt[j] = unlimbs(a[j]) < R_MOD
    return lt


class OsRng:
    """`OsRng` of the reference's call (sgx_dcap_verifier.rs:811): raw limbs from the operating system's CSPRNG.  What a production caller
    passes as `rng`; tests and benches pass a seeded numpy Generator so that proof bytes are reproducible."""

    def integers(self, low, high, size, dtype=np.uint64):
        assert low == 0 and high == 1 << 64 and dtype == np.uint64
        import os
        m = int(np.prod(size))
        return np.frombuffer(os.urandom(8 * m), dtype=np.uint64).reshape(size).copy()


def rand_fr_array(rng, n: int) -> np.ndarray:
    """n field elements UNIFORM in [0, r) by rejection (254-bit draws, accepted when below r: 75.6 % pass, rejected rows redrawn), as raw limbs.  A raw limb
    pattern v < r is the Montgomery form of v/R, and v -> v/R is a bijection of [0, r), so the elements are uniform whichever way the limbs
    are read.  This is the mirror's `Fr::random(&mut rng)` (blinding rows, the vanishing argument's random polynomial): `rng` is a numpy
    Generator (seeded: deterministic proofs for tests) or `OsRng()` (production)."""
    top_mask = np.uint64((1 << 62) - 1)
    out = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    out[:, 3] &= top_mask
    bad = np.nonzero(~_below_r(out))[0]
    while bad.size:                                                  # redraw only the rejected rows (24.4 % of a round)
        a = rng.integers(0, 1 << 64, size=(bad.size, 4), dtype=np.uint64)
        a[:, 3] &= top_mask
        out[bad] = a
        bad = bad[~_below_r(a)]
    return out
