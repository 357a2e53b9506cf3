"""halo2_proofs::arithmetic, MI355X edition.

Mirrors (halo2_proofs 0.2.0 @ zkwebauthn c254c75, Cargo.lock:1314-1327) src/arithmetic.rs:
    pub fn best_multiexp<C: CurveAffine>(coeffs: &[C::Scalar], bases: &[C]) -> C::Curve
    pub fn best_fft<G: Group>(a: &mut [G], omega: G::Scalar, log_n: u32)
reached from the reference through create_proof (circuits/src/sgx_dcap_verifier.rs:814-822).
Data are numpy uint64 arrays holding the Rust in-memory representation: Fr (n, 4), G1Affine (n, 8),
G1 (12,) — Montgomery limbs, little endian.
"""
from __future__ import annotations

import numpy as np

from ._lib import Backend, default_backend


class BasesHandle:
    """A base table resident on the GPU (params.g / params.g_lagrange of one SRS)."""

    def __init__(self, backend: Backend, bases):
        self.backend = backend
        if isinstance(bases, np.ndarray):
            self.n = bases.reshape(-1, 8).shape[0]
        else:
            self.n = bases[1]
        self.handle = backend.bases_register(bases)

    @classmethod
    def shared(cls, backend: Backend, owner: "BasesHandle") -> "BasesHandle":
        """a handle of `backend` (another context on the same GPU) onto the table `owner` registered — no second copy of the expanded table in HBM"""
        self = cls.__new__(cls)
        self.backend, self.n = backend, owner.n
        self.handle = backend.bases_share(owner.backend, owner.handle)
        return self

    def enable_runs(self):
        self.backend.bases_enable_runs(self.handle)
        return self

    def release(self):
        if self.handle:
            self.backend.bases_release(self.handle)
            self.handle = 0


def best_multiexp(coeffs, bases, backend: Backend | None = None) -> np.ndarray:
    """sum_i coeffs[i] * bases[i]  ->  G1 {x, y, z} (12 limbs), normalised (z = 1) or identity (0,0,0).

    `bases` is a BasesHandle (the production path: the table is uploaded once per SRS) or a raw
    (n, 8) array, which is registered for this call only.  Like the Rust function it asserts
    coeffs.len() == bases.len() for raw arrays; with a handle, len(coeffs) <= handle.n (a prefix of
    the table is used, as commit() does with shorter polynomials).
    """
    coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)
    if isinstance(bases, BasesHandle):
        assert coeffs.shape[0] <= bases.n
        return bases.backend.msm(bases.handle, coeffs)
    bases = np.ascontiguousarray(bases, dtype=np.uint64).reshape(-1, 8)
    assert coeffs.shape[0] == bases.shape[0], "best_multiexp: coeffs.len() != bases.len()"
    if coeffs.shape[0] == 0:
        return np.zeros(12, dtype=np.uint64)
    be = backend or default_backend()
    h = BasesHandle(be, bases)
    try:
        return be.msm(h.handle, coeffs)
    finally:
        h.release()


def best_fft(a: np.ndarray, omega, log_n: int, backend: Backend | None = None) -> None:
    """In place; natural order in and out: a[j] <- sum_i a[i] * omega^(i*j)."""
    be = backend or default_backend()
    assert a.shape[0] == 1 << log_n, "best_fft: a.len() != 1 << log_n"
    be.ntt(a, log_n, omega)


def best_multiexp_batch(columns, bases: BasesHandle) -> np.ndarray:
    """[best_multiexp(col, bases) for col in columns] in one device call (count, 12).

    halo2 commits the columns of a proof phase in a loop (src/plonk/prover.rs); the results are
    independent, so the shim hands the whole phase to the GPU at once and the latency-bound bucket
    reduction is paid once per phase instead of once per column."""
    return bases.backend.msm_batch(bases.handle, list(columns))


def eval_polynomial(poly: np.ndarray, point, backend: Backend | None = None) -> np.ndarray:
    """halo2_proofs::arithmetic::eval_polynomial(poly, point) -> Fr (host array in, 4 limbs out)."""
    be = backend or default_backend()
    poly = np.ascontiguousarray(poly, dtype=np.uint64).reshape(-1, 4)
    d = be.to_device(poly)
    out = be.eval_polynomial_batch_dev([d], poly.shape[0], [point])[0]
    d.free()
    return out


def kate_division(a: np.ndarray, b, backend: Backend | None = None) -> np.ndarray:
    """halo2_proofs::arithmetic::kate_division(a, b): quotient of a(X) by (X - b), len(a) - 1 coefficients."""
    be = backend or default_backend()
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    n = a.shape[0]
    d, q = be.to_device(a), be.alloc(max(n - 1, 1) * 32)
    be.kate_division_dev(d, n, b, q)
    out = q.download((n - 1, 4))
    d.free()
    q.free()
    return out
