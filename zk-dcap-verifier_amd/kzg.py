"""halo2_proofs::poly::kzg::commitment::ParamsKZG (commit side), MI355X edition.

Mirrors (halo2_proofs 0.2.0 @ zkwebauthn c254c75) src/poly/kzg/commitment.rs:
    commit_lagrange(poly, _blind) = best_multiexp(poly.values, g_lagrange[..n])
    commit(poly, _blind)          = best_multiexp(poly.values, g[..n])
(the Blind argument is ignored for KZG).  The two base tables are uploaded to
HBM once, when the params object is created — the analogue of gen_srs()/ParamsKZG::read at
circuits/src/sgx_dcap_verifier.rs:799.
"""
from __future__ import annotations

import numpy as np

from ._lib import Backend, default_backend
from .arithmetic import BasesHandle, best_multiexp, best_multiexp_batch


class ParamsKZG:
    def __init__(self, k: int, g: np.ndarray, g_lagrange: np.ndarray, backend: Backend | None = None):
        self.backend = backend or default_backend()
        self.k, self.n = k, 1 << k
        g = np.ascontiguousarray(g, dtype=np.uint64).reshape(-1, 8)
        g_lagrange = np.ascontiguousarray(g_lagrange, dtype=np.uint64).reshape(-1, 8)
        assert g.shape[0] == self.n and g_lagrange.shape[0] == self.n
        self.g = BasesHandle(self.backend, g)
        self.g_lagrange = BasesHandle(self.backend, g_lagrange)

    @classmethod
    def setup(cls, k: int, tau, backend: Backend | None = None) -> "ParamsKZG":
        """ParamsKZG::setup(k, rng) with the toxic waste `tau` given explicitly (TESTS / synthetic SRS only):
        g[i] = [tau^i] G  (n fixed-base multiplications), g_lagrange = EC-iFFT of g (g_to_lagrange) — both on the GPU."""
        be = backend or default_backend()
        n = 1 << k
        R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
        mont = lambda x: np.array([((x << 256) % R >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)
        powers = np.empty((n, 4), dtype=np.uint64)
        cur = 1
        for i in range(n):
            powers[i] = mont(cur)
            cur = cur * int(tau) % R
        ds, dg, dl = be.to_device(powers), be.alloc(n * 64), be.alloc(n * 64)
        be.g1_fixed_base_mul(ds, n, dg)
        omega_inv = pow(pow(7, (R - 1) >> k, R), R - 2, R)
        be.g1_ntt_dev(dg, k, mont(omega_inv), mont(pow(n, R - 2, R)), dl)
        self = cls.__new__(cls)
        self.backend, self.k, self.n = be, k, n
        self.g_host, self.g_lagrange_host = dg.download((n, 8)), dl.download((n, 8))
        self.g, self.g_lagrange = BasesHandle(be, (dg, n)), BasesHandle(be, (dl, n))
        for d in (ds, dg, dl):
            d.free()
        return self

    def commit(self, poly: np.ndarray) -> np.ndarray:
        poly = np.asarray(poly, dtype=np.uint64).reshape(-1, 4)
        assert poly.shape[0] <= self.n
        return best_multiexp(poly, self.g)

    def commit_lagrange(self, poly: np.ndarray) -> np.ndarray:
        poly = np.asarray(poly, dtype=np.uint64).reshape(-1, 4)
        assert poly.shape[0] == self.n
        return best_multiexp(poly, self.g_lagrange)

    def commit_lagrange_batch(self, polys) -> np.ndarray:
        return best_multiexp_batch(polys, self.g_lagrange)

    def commit_batch(self, polys) -> np.ndarray:
        return best_multiexp_batch(polys, self.g)

    def release(self):
        self.g.release()
        self.g_lagrange.release()
