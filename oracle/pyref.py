"""TEST INFRASTRUCTURE ONLY — pure-Python big-integer model of the BN254 hot path.

This file is the *definition-level* oracle: it states what MSM, NTT, the halo2
EvaluationDomain operations and the h(X) numerator ARE, as mathematics over
Python integers.  It is used to (a) pin the C restatement in ``oracle/*.c`` on
small sizes and (b) generate the committed fixtures under ``tests/golden/``.
Nothing in the product path (``zk-dcap-verifier_amd/``) may import it.

PARITY STATUS: **parity unpinned by the reference**.  The reference
(/root/reference) only *calls* the prover (circuits/src/sgx_dcap_verifier.rs:799-822,
crates/p256-ecdsa/src/base.rs:134,145,193-212); the arithmetic lives in
un-vendored crates (halo2_proofs 0.2.0 @ zkwebauthn c254c75, Cargo.lock:1314-1327;
halo2curves 0.3.1 @ bdb2e66, Cargo.lock:1329-1344) and no reference test holds
an MSM / NTT / h(X) value.  What *is* pinned here:
  * the field/curve constants (moduli, R, ROOT_OF_UNITY, ZETA, DELTA …) are
    re-derived from first principles and checked against SURVEY.md App. A;
  * the only proof bytes the reference ships (bin/assets/proof.bin, used by
    bin/src/main.rs:269-279) decode to 13+2 points ON y^2 = x^3 + 3 over this
    Fq and 32 scalars < this r  (tests/test_oracle_golden.py);
  * MSM / NTT outputs are canonical mathematical objects, so any correct
    implementation agrees after normalisation.
"""
from __future__ import annotations

# ----------------------------------------------------------------------------
# Constants (halo2curves::bn256 — names as in that crate; values recomputed)
# ----------------------------------------------------------------------------
P = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47  # Fq
R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001  # Fr
B_COEFF = 3
G1_GEN = (1, 2)
S = 28                        # two-adicity of Fr
MULT_GEN = 7                  # Fr::MULTIPLICATIVE_GENERATOR
MONT_BITS = 256


def _root_of_unity() -> int:
    return pow(MULT_GEN, (R - 1) >> S, R)


ROOT_OF_UNITY = _root_of_unity()
ROOT_OF_UNITY_INV = pow(ROOT_OF_UNITY, R - 2, R)
DELTA = pow(MULT_GEN, 1 << S, R)
ZETA = pow(MULT_GEN, (R - 1) // 3, R)
TWO_INV = pow(2, R - 2, R)


def mont_r(mod: int) -> int:
    return (1 << MONT_BITS) % mod


def to_mont(x: int, mod: int) -> int:
    return (x << MONT_BITS) % mod


def from_mont(x: int, mod: int) -> int:
    return (x * pow(1 << MONT_BITS, -1, mod)) % mod


def mont_inv64(mod: int) -> int:
    """-mod^{-1} mod 2^64 (the INV constant of halo2curves' field macros)."""
    return (-pow(mod, -1, 1 << 64)) % (1 << 64)


def limbs64(x: int, n: int = 4):
    return [(x >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)]


def limbs32(x: int, n: int = 8):
    return [(x >> (32 * i)) & 0xFFFFFFFF for i in range(n)]


def from_limbs64(l):
    return sum(int(v) << (64 * i) for i, v in enumerate(l))


def omega(k: int) -> int:
    """Primitive 2^k-th root of unity as EvaluationDomain derives it."""
    assert 0 <= k <= S
    return pow(ROOT_OF_UNITY, 1 << (S - k), R)


# ----------------------------------------------------------------------------
# G1 (y^2 = x^3 + 3 over Fq), affine with None = identity
# ----------------------------------------------------------------------------
def g1_is_on_curve(pt) -> bool:
    if pt is None:
        return True
    x, y = pt
    return (y * y - x * x * x - B_COEFF) % P == 0


def g1_neg(pt):
    if pt is None:
        return None
    return (pt[0], (-pt[1]) % P)


def g1_add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    x1, y1 = a
    x2, y2 = b
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return None
        lam = (3 * x1 * x1) * pow(2 * y1, -1, P) % P
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    y3 = (lam * (x1 - x3) - y1) % P
    return (x3, y3)


def g1_mul(pt, k: int):
    k %= R
    acc = None
    add = pt
    while k:
        if k & 1:
            acc = g1_add(acc, add)
        add = g1_add(add, add)
        k >>= 1
    return acc


def msm_naive(scalars, points):
    """Definition: sum_i scalars[i] * points[i]."""
    acc = None
    for s, p in zip(scalars, points):
        acc = g1_add(acc, g1_mul(p, s))
    return acc


def g1_decompress_x(x: int):
    """Return the two candidate y for x (or None if x is not on the curve)."""
    rhs = (x * x * x + B_COEFF) % P
    y = pow(rhs, (P + 1) // 4, P)  # P = 3 mod 4
    if y * y % P != rhs:
        return None
    return (y, (-y) % P)


# ----------------------------------------------------------------------------
# NTT — definition: out[j] = sum_i a[i] * w^(i*j), natural order
# ----------------------------------------------------------------------------
def ntt_definition(a, w):
    n = len(a)
    return [sum(a[i] * pow(w, i * j, R) for i in range(n)) % R for j in range(n)]


def ntt_fast(a, w):
    """Recursive radix-2, same result as ntt_definition (used for n up to 2^12)."""
    n = len(a)
    if n == 1:
        return list(a)
    even = ntt_fast(a[0::2], w * w % R)
    odd = ntt_fast(a[1::2], w * w % R)
    out = [0] * n
    t = 1
    for j in range(n // 2):
        u = odd[j] * t % R
        out[j] = (even[j] + u) % R
        out[j + n // 2] = (even[j] - u) % R
        t = t * w % R
    return out


def poly_eval(coeffs, x):
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % R
    return acc


# ----------------------------------------------------------------------------
# EvaluationDomain (halo2_proofs poly/domain.rs semantics)
# ----------------------------------------------------------------------------
class Domain:
    def __init__(self, j: int, k: int):
        self.k = k
        self.n = 1 << k
        self.quotient_poly_degree = j - 1
        ek = k
        while (1 << ek) < self.n * self.quotient_poly_degree:
            ek += 1
        self.extended_k = ek
        self.extended_omega = omega(ek)
        self.omega = pow(self.extended_omega, 1 << (ek - k), R)
        assert self.omega == omega(k)
        self.omega_inv = pow(self.omega, R - 2, R)
        self.extended_omega_inv = pow(self.extended_omega, R - 2, R)
        self.g_coset = ZETA
        self.g_coset_inv = ZETA * ZETA % R
        self.ifft_divisor = pow(self.n, R - 2, R)
        self.extended_ifft_divisor = pow(1 << ek, R - 2, R)
        # t_evaluations[i] = (ZETA^n * (ext_omega^n)^i - 1)^-1, i < 2^(ek-k)
        zn = pow(ZETA, self.n, R)
        won = pow(self.extended_omega, self.n, R)
        self.t_evaluations = []
        cur = zn
        for _ in range(1 << (ek - k)):
            self.t_evaluations.append(pow((cur - 1) % R, R - 2, R))
            cur = cur * won % R

    def lagrange_to_coeff(self, a):
        out = ntt_fast(a, self.omega_inv)
        return [x * self.ifft_divisor % R for x in out]

    def coeff_to_lagrange(self, a):
        return ntt_fast(a, self.omega)

    def coeff_to_extended(self, a):
        assert len(a) == self.n
        z = [1, ZETA, ZETA * ZETA % R]
        b = [a[i] * z[i % 3] % R for i in range(self.n)]
        b += [0] * ((1 << self.extended_k) - self.n)
        return ntt_fast(b, self.extended_omega)

    def extended_to_coeff(self, a):
        assert len(a) == 1 << self.extended_k
        b = ntt_fast(a, self.extended_omega_inv)
        zi = [1, ZETA * ZETA % R, ZETA]  # ZETA^-(i mod 3)
        b = [b[i] * self.extended_ifft_divisor % R * zi[i % 3] % R for i in range(len(b))]
        return b[: self.n * self.quotient_poly_degree]

    def divide_by_vanishing_poly(self, a):
        m = len(self.t_evaluations)
        return [a[i] * self.t_evaluations[i % m] % R for i in range(len(a))]

    def extended_point(self, idx: int) -> int:
        """The idx-th point of the extended coset: ZETA * ext_omega^idx."""
        return ZETA * pow(self.extended_omega, idx, R) % R


if __name__ == "__main__":
    print("p   =", hex(P))
    print("r   =", hex(R))
    print("ROOT_OF_UNITY =", hex(ROOT_OF_UNITY))
    print("ZETA =", hex(ZETA), "DELTA =", hex(DELTA))
