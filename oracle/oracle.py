"""TEST INFRASTRUCTURE ONLY — ctypes loader for oracle/build/liboracle.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module (see bn254_oracle.c header; parity unpinned by the reference).
Field elements travel as numpy uint64 arrays of shape (n, 4) (Montgomery limbs,
little endian) — the in-memory layout of halo2curves bn256::{Fr,Fq}.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "build", "liboracle.so")


def build(force: bool = False) -> str:
    src = [os.path.join(HERE, f) for f in ("bn254_oracle.c", "evaluate_h_oracle.inc", "bn254_consts.h")]
    if force or not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in src if os.path.exists(s)):
        subprocess.check_call(["make", "-C", HERE, "-s"])
    return SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_domain_new.restype = C.c_int
        _lib.orc_evaluate_h.restype = C.c_int
        _lib.orc_g1_is_on_curve.restype = C.c_int
        _lib.orc_lookup_permute.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _fe(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    assert a.shape[-1] == 4
    return a


def _vec2(name, a, b):
    a, b = _fe(a), _fe(b)
    out = np.empty_like(a)
    getattr(lib(), name)(_p(a), _p(b), _p(out), C.c_size_t(a.size // 4))
    return out


def _vec1(name, a):
    a = _fe(a)
    out = np.empty_like(a)
    getattr(lib(), name)(_p(a), _p(out), C.c_size_t(a.size // 4))
    return out


def fr_mul(a, b): return _vec2("orc_fr_mul_vec", a, b)
def fr_add(a, b): return _vec2("orc_fr_add_vec", a, b)
def fr_sub(a, b): return _vec2("orc_fr_sub_vec", a, b)
def fq_mul(a, b): return _vec2("orc_fq_mul_vec", a, b)
def fq_add(a, b): return _vec2("orc_fq_add_vec", a, b)
def fq_sub(a, b): return _vec2("orc_fq_sub_vec", a, b)
def fr_to_mont(a): return _vec1("orc_fr_to_mont", a)
def fr_from_mont(a): return _vec1("orc_fr_from_mont", a)
def fq_to_mont(a): return _vec1("orc_fq_to_mont", a)
def fq_from_mont(a): return _vec1("orc_fq_from_mont", a)
def fr_inv(a): return _vec1("orc_fr_inv", a)
def fq_inv(a): return _vec1("orc_fq_inv", a)


# ---- integer <-> limb helpers (test convenience) ---------------------------
def ints_to_limbs(vals) -> np.ndarray:
    out = np.empty((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        for j in range(4):
            out[i, j] = (int(v) >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
    return out


def limbs_to_ints(a) -> list:
    a = np.asarray(a, dtype=np.uint64).reshape(-1, 4)
    return [sum(int(a[i, j]) << (64 * j) for j in range(4)) for i in range(a.shape[0])]


def fr_from_ints(vals): return fr_to_mont(ints_to_limbs(vals))
def fr_to_ints(a): return limbs_to_ints(fr_from_mont(np.asarray(a).reshape(-1, 4)))
def fq_from_ints(vals): return fq_to_mont(ints_to_limbs(vals))
def fq_to_ints(a): return limbs_to_ints(fq_from_mont(np.asarray(a).reshape(-1, 4)))


# ---- G1 ---------------------------------------------------------------------
def g1_generator() -> np.ndarray:
    out = np.zeros((1, 8), dtype=np.uint64)
    lib().orc_g1_generator(_p(out))
    return out


def g1_affine_from_ints(pts) -> np.ndarray:
    """pts: list of (x, y) or None -> (n, 8) uint64 Montgomery affine."""
    xs = [0 if p is None else p[0] for p in pts]
    ys = [0 if p is None else p[1] for p in pts]
    return np.concatenate([fq_from_ints(xs), fq_from_ints(ys)], axis=1)


def g1_affine_to_ints(a) -> list:
    a = np.asarray(a, dtype=np.uint64).reshape(-1, 8)
    xs = fq_to_ints(a[:, :4].copy())
    ys = fq_to_ints(a[:, 4:].copy())
    return [None if (x == 0 and y == 0) else (x, y) for x, y in zip(xs, ys)]


def g1_to_affine(jac) -> np.ndarray:
    jac = np.ascontiguousarray(jac, dtype=np.uint64).reshape(-1, 12)
    out = np.empty((jac.shape[0], 8), dtype=np.uint64)
    lib().orc_g1_to_affine(_p(jac), _p(out), C.c_size_t(jac.shape[0]))
    return out


def g1_mul(p_affine, k_mont) -> np.ndarray:
    p_affine = np.ascontiguousarray(p_affine, dtype=np.uint64).reshape(8)
    k_mont = _fe(np.asarray(k_mont).reshape(4))
    out = np.empty(12, dtype=np.uint64)
    lib().orc_g1_mul(_p(p_affine), _p(k_mont), _p(out))
    return out


def g1_add(a_jac, b_jac) -> np.ndarray:
    a = np.ascontiguousarray(a_jac, dtype=np.uint64).reshape(12)
    b = np.ascontiguousarray(b_jac, dtype=np.uint64).reshape(12)
    out = np.empty(12, dtype=np.uint64)
    lib().orc_g1_add(_p(a), _p(b), _p(out))
    return out


def g1_is_on_curve(p_affine) -> bool:
    p = np.ascontiguousarray(p_affine, dtype=np.uint64).reshape(8)
    return bool(lib().orc_g1_is_on_curve(_p(p)))


def gen_bases_arith(a0: int, delta: int, n: int, threads: int = 8) -> np.ndarray:
    a0m, dm = fr_from_ints([a0]), fr_from_ints([delta])
    out = np.empty((n, 8), dtype=np.uint64)
    lib().orc_gen_bases_arith(_p(a0m), _p(dm), C.c_size_t(n), C.c_int(threads), _p(out))
    return out


def best_multiexp(coeffs, bases, threads: int = 8) -> np.ndarray:
    """halo2 best_multiexp semantics; returns the Jacobian result (12 limbs)."""
    coeffs = _fe(coeffs)
    bases = np.ascontiguousarray(bases, dtype=np.uint64).reshape(-1, 8)
    assert coeffs.shape[0] == bases.shape[0]
    out = np.empty(12, dtype=np.uint64)
    lib().orc_best_multiexp(_p(coeffs), _p(bases), C.c_size_t(coeffs.shape[0]), C.c_int(threads), _p(out))
    return out


def best_fft(a, omega_mont, log_n: int, threads: int = 8) -> np.ndarray:
    a = _fe(a).copy()
    assert a.shape[0] == 1 << log_n
    w = _fe(np.asarray(omega_mont).reshape(4))
    lib().orc_best_fft(_p(a), _p(w), C.c_uint32(log_n), C.c_int(threads))
    return a


class Domain:
    """EvaluationDomain::new(j, k)."""

    class _S(C.Structure):
        _fields_ = [("k", C.c_uint32), ("extended_k", C.c_uint32), ("quotient_poly_degree", C.c_uint32), ("n_t", C.c_uint32),
                    ("omega", C.c_uint64 * 4), ("omega_inv", C.c_uint64 * 4), ("extended_omega", C.c_uint64 * 4),
                    ("extended_omega_inv", C.c_uint64 * 4), ("ifft_divisor", C.c_uint64 * 4),
                    ("extended_ifft_divisor", C.c_uint64 * 4), ("t_evaluations", (C.c_uint64 * 4) * 64)]

    def __init__(self, j: int, k: int):
        self.s = Domain._S()
        rc = lib().orc_domain_new(C.byref(self.s), C.c_uint32(j), C.c_uint32(k))
        if rc:
            raise ValueError("bad domain")
        self.k, self.extended_k = self.s.k, self.s.extended_k
        self.n, self.extended_n = 1 << self.k, 1 << self.extended_k

    def _fe(self, name):
        return np.array(list(getattr(self.s, name)), dtype=np.uint64)

    @property
    def omega(self): return self._fe("omega")
    @property
    def omega_inv(self): return self._fe("omega_inv")
    @property
    def extended_omega(self): return self._fe("extended_omega")
    @property
    def extended_omega_inv(self): return self._fe("extended_omega_inv")

    def lagrange_to_coeff(self, a, threads=8):
        a = _fe(a).copy(); lib().orc_lagrange_to_coeff(C.byref(self.s), _p(a), C.c_int(threads)); return a

    def coeff_to_lagrange(self, a, threads=8):
        a = _fe(a).copy(); lib().orc_coeff_to_lagrange(C.byref(self.s), _p(a), C.c_int(threads)); return a

    def coeff_to_extended(self, a, threads=8):
        a = _fe(a); out = np.empty((self.extended_n, 4), dtype=np.uint64)
        lib().orc_coeff_to_extended(C.byref(self.s), _p(a), _p(out), C.c_int(threads)); return out

    def extended_to_coeff(self, a, threads=8):
        a = _fe(a).copy(); lib().orc_extended_to_coeff(C.byref(self.s), _p(a), C.c_int(threads))
        return a[: self.n * self.s.quotient_poly_degree]

    def divide_by_vanishing_poly(self, a):
        a = _fe(a).copy(); lib().orc_divide_by_vanishing_poly(C.byref(self.s), _p(a)); return a


def eval_polynomial(poly, x_mont) -> np.ndarray:
    poly = _fe(poly); x = _fe(np.asarray(x_mont).reshape(4)); out = np.empty(4, dtype=np.uint64)
    lib().orc_eval_polynomial(_p(poly), C.c_size_t(poly.shape[0]), _p(x), _p(out)); return out


def _ptr_array(cols):
    cols = [np.ascontiguousarray(c, dtype=np.uint64) for c in cols]
    arr = (C.c_void_p * max(1, len(cols)))(*[c.ctypes.data for c in cols])
    return arr, cols


def evaluate_h(blob: bytes, fixed, advice, instance, l0, l_last, l_active_row, perm_cosets, perm_products,
               lk_product, lk_input, lk_table, challenges, beta, gamma, theta, y, extended_n: int, threads: int = 8):
    """All column arguments are lists of (extended_n, 4) coset arrays."""
    keep = []
    def pa(cols):
        arr, c = _ptr_array(cols); keep.append(c); return arr
    blob_arr = np.frombuffer(blob, dtype=np.uint32).copy()
    values = np.zeros((extended_n, 4), dtype=np.uint64)
    ch = _fe(np.asarray(challenges, dtype=np.uint64).reshape(-1, 4)) if len(challenges) else np.zeros((1, 4), dtype=np.uint64)
    sc = [_fe(np.asarray(v).reshape(4)) for v in (beta, gamma, theta, y)]
    rc = lib().orc_evaluate_h(_p(blob_arr), pa(fixed), pa(advice), pa(instance), _p(_fe(l0)), _p(_fe(l_last)), _p(_fe(l_active_row)),
                              pa(perm_cosets), pa(perm_products), C.c_uint32(len(perm_products)),
                              pa(lk_product), pa(lk_input), pa(lk_table), _p(ch), _p(sc[0]), _p(sc[1]), _p(sc[2]), _p(sc[3]),
                              _p(values), C.c_int(threads))
    if rc:
        raise ValueError("bad program blob")
    return values


def permutation_product(values, sigmas, k, beta, gamma, delta_start, last_z, blinding):
    """One column set of permutation::Argument::commit -> (z (n,4), last_z (4,))."""
    keep = []
    def pa(cols):
        arr, c = _ptr_array(cols); keep.append(c); return arr
    n = 1 << k
    z = np.empty((n, 4), dtype=np.uint64); last = np.empty(4, dtype=np.uint64)
    bl = np.ascontiguousarray(np.asarray(blinding, dtype=np.uint64).reshape(-1, 4))
    sc = [_fe(np.asarray(v).reshape(4)) for v in (beta, gamma, delta_start, last_z)]
    lib().orc_permutation_product(pa(values), pa(sigmas), C.c_size_t(len(values)), C.c_uint32(k), _p(sc[0]), _p(sc[1]), _p(sc[2]), _p(sc[3]),
                                  _p(bl), C.c_uint32(bl.shape[0]), _p(z), _p(last))
    return z, last


def lookup_product(cin, ctab, pin, ptab, k, beta, gamma, blinding):
    n = 1 << k
    z = np.empty((n, 4), dtype=np.uint64)
    bl = np.ascontiguousarray(np.asarray(blinding, dtype=np.uint64).reshape(-1, 4))
    sc = [_fe(np.asarray(v).reshape(4)) for v in (beta, gamma)]
    lib().orc_lookup_product(_p(_fe(cin)), _p(_fe(ctab)), _p(_fe(pin)), _p(_fe(ptab)), C.c_uint32(k), _p(sc[0]), _p(sc[1]), _p(bl),
                             C.c_uint32(bl.shape[0]), _p(z))
    return z


def kate_division(a, b_mont) -> np.ndarray:
    a = _fe(a); b = _fe(np.asarray(b_mont).reshape(4))
    q = np.empty((a.shape[0] - 1, 4), dtype=np.uint64)
    lib().orc_kate_division(_p(a), C.c_size_t(a.shape[0]), _p(b), _p(q))
    return q


def lookup_permute(inp, table, k, blinding_factors, blind_in, blind_tab):
    """permute_expression_pair -> (permuted_input, permuted_table); raises ValueError for ConstraintSystemFailure."""
    n = 1 << k
    oi, ot = np.empty((n, 4), dtype=np.uint64), np.empty((n, 4), dtype=np.uint64)
    rc = lib().orc_lookup_permute(_p(_fe(inp)), _p(_fe(table)), C.c_uint32(k), C.c_uint32(blinding_factors), _p(_fe(blind_in)), _p(_fe(blind_tab)), _p(oi), _p(ot))
    if rc:
        raise ValueError("ConstraintSystemFailure: input value not in table")
    return oi, ot


def g1_fft(points_affine, log_n: int, omega_mont, scale_mont=None) -> np.ndarray:
    pts = np.ascontiguousarray(points_affine, dtype=np.uint64).reshape(1 << log_n, 8)
    w = _fe(np.asarray(omega_mont).reshape(4))
    out = np.empty_like(pts)
    sc = _fe(np.asarray(scale_mont).reshape(4)) if scale_mont is not None else None
    lib().orc_g1_fft(_p(pts), C.c_uint32(log_n), _p(w), _p(sc) if sc is not None else None, _p(out))
    return out
