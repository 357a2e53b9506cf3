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
        self.g_host, self.g_lagrange_host = g, g_lagrange
        self.g = BasesHandle(self.backend, g)
        self.g_lagrange = BasesHandle(self.backend, g_lagrange).enable_runs()   # Lagrange columns have runs (sorted lookup inputs, constant regions): see zk_bases_enable_runs

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
        self.g, self.g_lagrange = BasesHandle(be, (dg, n)), BasesHandle(be, (dl, n)).enable_runs()
        for d in (ds, dg, dl):
            d.free()
        return self

    @classmethod
    def shared_with(cls, other: "ParamsKZG", backend: Backend) -> "ParamsKZG":
        """The same SRS for another context on the same GPU (one context per host thread that proves concurrently): both window-expanded tables
        are SHARED with `other` (zk_bases_share) — one copy per process, 2 x 512 MiB at k = 19 instead of that per context."""
        self = cls.__new__(cls)
        self.backend, self.k, self.n = backend, other.k, other.n
        self.g_host, self.g_lagrange_host = other.g_host, other.g_lagrange_host
        self.g, self.g_lagrange = BasesHandle.shared(backend, other.g), BasesHandle.shared(backend, other.g_lagrange)
        return self

    # -- ParamsKZG::{write, read}: k (u32 LE) | n compressed g | n compressed g_lagrange | g2 | s_g2  [3P-MEM: SURVEY §8f n3, App. C.7] ----------
    def write(self, g2: bytes = bytes(64), s_g2: bytes = bytes(64), sign_bit: int = 255) -> bytes:
        """The params/kzg_bn254_{k}.srs byte stream.  The G2 points are opaque 64-byte blobs here (the prover never touches them)."""
        be, n = self.backend, self.n
        out = [int(self.k).to_bytes(4, "little")]
        for host in (self.g_host, self.g_lagrange_host):
            d, b = be.to_device(np.ascontiguousarray(host, dtype=np.uint64)), be.alloc(n * 32)
            be.g1_compress_dev(d, n, sign_bit, b)
            out.append(b.download((n, 4)).tobytes())
            d.free(); b.free()
        assert len(g2) == 64 and len(s_g2) == 64
        return b"".join(out) + bytes(g2) + bytes(s_g2)

    @classmethod
    def read(cls, data: bytes, backend: Backend | None = None, sign_bit: int = 255) -> "ParamsKZG":
        """ParamsKZG::read: decompress both tables on the GPU (one square root per point) and register them.  Raises ZkError when an
        encoding is not a curve point.  `.g2` / `.s_g2` keep the 64-byte tails untouched."""
        be = backend or default_backend()
        k = int.from_bytes(data[:4], "little")
        n = 1 << k
        if k > 27 or len(data) != 4 + 2 * n * 32 + 128:
            raise ValueError("not a kzg_bn254 SRS stream (k out of range or length mismatch)")
        hosts = []
        for t in range(2):
            raw = np.frombuffer(data, dtype=np.uint64, count=n * 4, offset=4 + t * n * 32).reshape(n, 4)
            b, d = be.to_device(raw), be.alloc(n * 64)
            be.g1_decompress_dev(b, n, sign_bit, d)
            hosts.append(d.download((n, 8)))
            b.free(); d.free()
        self = cls(k, hosts[0], hosts[1], backend=be)
        self.g_host, self.g_lagrange_host = hosts
        self.g2, self.s_g2 = data[-128:-64], data[-64:]
        return self

    # -- multi-GPU: base tables sharded by index range, partial points combined after an all-gather --------------------------------
    world, rank, lo, n_loc, all_gather = 1, 0, 0, None, None
    coset_exchange = None
    quotient_by_cosets = False       # world == 1 only: take the coset-by-coset quotient path anyway (tests; the result is the same proof)

    def by_cosets(self) -> bool:
        return self.world > 1 or self.quotient_by_cosets

    def quotient_parts(self, n_cosets: int) -> int:
        """how many ranks share one coset: 1 while there are at least as many cosets as ranks; with more ranks (8 GPUs at extended_k = k + 2) every coset's
        rows are cut into world / n_cosets aligned slices (the ranks of a coset each run the size-n NTTs of that coset and evaluate their slice of its rows)"""
        if self.world <= n_cosets or self.world % n_cosets:
            return 1
        parts = self.world // n_cosets
        return parts if parts & (parts - 1) == 0 and self.n // parts >= 1 and self.n % parts == 0 else 1

    def my_units(self, n_cosets: int) -> list:
        """The extended domain is 2^(extended_k - k) interleaved cosets of the 2^k domain; evaluate_h never mixes them (rotations stay inside
        a coset), so the quotient shards by (coset, slice of its rows): unit u = coset u // parts, rows [(u % parts) * n / parts, +n / parts);
        rank r evaluates units [r * slots, (r + 1) * slots), slots = ceil(units / world).  -> [(coset, row_lo, row_count)]"""
        parts = self.quotient_parts(n_cosets)
        units, rows = n_cosets * parts, self.n // parts
        slots = -(-units // self.world)
        return [(u // parts, (u % parts) * rows, rows) for u in range(self.rank * slots, (self.rank + 1) * slots) if u < units]

    def my_cosets(self, n_cosets: int) -> list:
        """the cosets this rank needs the proving key's columns on"""
        return sorted({j for j, _, _ in self.my_units(n_cosets)})

    def gather_cosets(self, mine: np.ndarray, n_cosets: int) -> np.ndarray:
        """mine: (slots, rows, 4) numerator values of this rank's units (unused slots zero) -> (n_cosets, n, 4), every rank's, in coset order."""
        if self.world == 1:
            return mine.reshape(-1, self.n, 4)[:n_cosets]
        return self.all_gather(mine).reshape(-1, 4)[:n_cosets * self.n].reshape(n_cosets, self.n, 4)

    @classmethod
    def sharded(cls, k: int, g: np.ndarray, g_lagrange: np.ndarray, rank: int, world: int, all_gather, backend: Backend | None = None,
                coset_exchange=None) -> "ParamsKZG":
        """Rank `rank` of `world` holds bases [rank * n / world, (rank + 1) * n / world) of both tables (window-expanded in ITS HBM only).
        `all_gather(partials: (count, 16) uint64) -> (world, count, 16)` exchanges the 128-byte XYZZ partial sums — one RCCL all_gather per
        commitment phase; EC addition is not an RCCL reduction, so every rank adds the world partial points itself.
        `coset_exchange(nbytes) -> (send_ptr, recv_ptr, run)`, optional: device memory for the quotient's bulk all-gather (the caller owns it, e.g.
        two torch tensors); `run()` gathers every rank's nbytes at send_ptr into recv_ptr[rank * nbytes ...] and returns when the data is there
        (it must order itself against this library's stream: Backend.sync() before, a device synchronise after).  Without it the numerators
        travel through `all_gather` as host arrays."""
        be = backend or default_backend()
        n = 1 << k
        assert n % world == 0 and 0 <= rank < world
        g = np.ascontiguousarray(g, dtype=np.uint64).reshape(n, 8)
        g_lagrange = np.ascontiguousarray(g_lagrange, dtype=np.uint64).reshape(n, 8)
        self = cls.__new__(cls)
        self.backend, self.k, self.n = be, k, n
        self.world, self.rank, self.n_loc, self.all_gather = world, rank, n // world, all_gather
        self.lo = rank * self.n_loc
        self.coset_exchange = coset_exchange
        self.g_host, self.g_lagrange_host = g, g_lagrange
        self.g = BasesHandle(be, np.ascontiguousarray(g[self.lo:self.lo + self.n_loc]))
        self.g_lagrange = BasesHandle(be, np.ascontiguousarray(g_lagrange[self.lo:self.lo + self.n_loc])).enable_runs()
        return self

    def commit_columns(self, which: str, cols) -> np.ndarray:
        """[commit(col) for col in cols] against `g` or `g_lagrange` for DEVICE columns of n scalars -> (count, 12) normalised points.
        One device call per rank; with sharded tables each rank multiplies its index range and the partial points are all-gathered."""
        from ._lib import _dptr
        h = self.g if which == "g" else self.g_lagrange
        be = self.backend
        if not len(cols):
            return np.zeros((0, 12), dtype=np.uint64)
        if self.world == 1:
            return be.msm_batch(h.handle, list(cols), self.n)
        part = be.msm_batch_partial(h.handle, [_dptr(c) + self.lo * 32 for c in cols], self.n_loc)
        return be.g1_sum_xyzz_batch(self.all_gather(part))

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
