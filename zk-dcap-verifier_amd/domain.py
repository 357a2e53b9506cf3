"""halo2_proofs::poly::EvaluationDomain, MI355X edition.

Mirrors (halo2_proofs 0.2.0 @ zkwebauthn c254c75) src/poly/domain.rs: EvaluationDomain::new(j, k),
lagrange_to_coeff, coeff_to_extended, extended_to_coeff, divide_by_vanishing_poly — the NTT wrappers
create_proof calls (reference call site circuits/src/sgx_dcap_verifier.rs:814-822).  The constants
(extended_k rule, omega from ROOT_OF_UNITY, ZETA coset, t_evaluations) follow SURVEY.md App. C.3.
Host arrays in, host arrays out; `*_dev` variants keep data in HBM.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import Backend, default_backend


class EvaluationDomain:
    def __init__(self, j: int, k: int, backend: Backend | None = None):
        self.backend = backend or default_backend()
        self.k = k
        self.n = 1 << k
        self.quotient_poly_degree = j - 1
        ek = k
        while (1 << ek) < self.n * self.quotient_poly_degree:
            ek += 1
        if ek > 28:
            raise ValueError("extended_k exceeds the two-adicity of Fr")
        self.extended_k = ek
        self.extended_n = 1 << ek

    def extended_len(self) -> int:
        return self.extended_n

    # host-buffer forms -------------------------------------------------------------------------
    def lagrange_to_coeff(self, a: np.ndarray) -> np.ndarray:
        a = np.ascontiguousarray(a, dtype=np.uint64).reshape(self.n, 4).copy()
        be = self.backend
        be._ck(be.lib.zk_lagrange_to_coeff(be.ctx, a.ctypes.data_as(C.c_void_p), C.c_uint32(self.k)))
        return a

    def coeff_to_extended(self, a: np.ndarray) -> np.ndarray:
        a = np.ascontiguousarray(a, dtype=np.uint64).reshape(self.n, 4)
        out = np.empty((self.extended_n, 4), dtype=np.uint64)
        be = self.backend
        be._ck(be.lib.zk_coeff_to_extended(be.ctx, a.ctypes.data_as(C.c_void_p), C.c_uint32(self.k), C.c_uint32(self.extended_k),
                                           out.ctypes.data_as(C.c_void_p)))
        return out

    def extended_to_coeff(self, a: np.ndarray) -> np.ndarray:
        a = np.ascontiguousarray(a, dtype=np.uint64).reshape(self.extended_n, 4).copy()
        be = self.backend
        be._ck(be.lib.zk_extended_to_coeff(be.ctx, a.ctypes.data_as(C.c_void_p), C.c_uint32(self.k), C.c_uint32(self.extended_k)))
        return a[: self.n * self.quotient_poly_degree]

    def divide_by_vanishing_poly(self, a: np.ndarray) -> np.ndarray:
        d = self.backend.to_device(np.ascontiguousarray(a, dtype=np.uint64).reshape(self.extended_n, 4))
        self.backend.divide_by_vanishing_poly_dev(d, self.k, self.extended_k)
        out = d.download((self.extended_n, 4))
        d.free()
        return out

    # device-resident forms ---------------------------------------------------------------------
    def lagrange_to_coeff_dev(self, a_dev): self.backend.lagrange_to_coeff_dev(a_dev, self.k)
    def lagrange_to_coeff_batch_dev(self, cols): self.backend.lagrange_to_coeff_batch_dev(cols, self.k)
    def coeff_to_extended_batch_dev(self, coeffs, outs): self.backend.coeff_to_extended_batch_dev(coeffs, outs, self.k, self.extended_k)
    def coeff_to_extended_dev(self, coeff_dev, out_dev): self.backend.coeff_to_extended_dev(coeff_dev, self.k, self.extended_k, out_dev)
    def extended_to_coeff_dev(self, a_dev): self.backend.extended_to_coeff_dev(a_dev, self.k, self.extended_k)
    def divide_by_vanishing_poly_dev(self, a_dev): self.backend.divide_by_vanishing_poly_dev(a_dev, self.k, self.extended_k)
