"""halo2_proofs::poly::kzg::multiopen::ProverSHPLONK, MI355X edition.

Mirrors (halo2_proofs 0.2.0 @ zkwebauthn c254c75, Cargo.lock:1314-1327) src/poly/kzg/multiopen/shplonk{.rs,/prover.rs} —
the multi-open argument the reference instantiates as `ProverSHPLONK<Bn256>` in its create_proof call
(circuits/src/sgx_dcap_verifier.rs:814-822).  The polynomial work — linear combinations over n coefficients,
division by (X - point), the two commitments — runs on the GPU (zk_fr_lincomb_dev, zk_kate_division_dev, zk_msm_dev);
the set bookkeeping and the O(1)-size interpolations stay on the host as in the reference.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

from ..fields import R_MOD, fr_mont, fr_mont_array, g1_affine_ints


class ProverQuery:
    """ProverQuery { point, poly }: `poly` is a device buffer of n coefficients; `eval` = poly(point) (already written to the transcript)."""
    __slots__ = ("poly", "point", "eval")

    def __init__(self, poly, point: int, eval_: int):
        self.poly, self.point, self.eval = poly, point % R_MOD, eval_ % R_MOD


def construct_intermediate_sets(queries: Sequence, key=lambda q: id(q.poly)):
    """shplonk.rs construct_intermediate_sets: group commitments by their set of opening points.
    Returns (rotation_sets, super_point_set) with rotation_sets = [(points sorted, [(query-of-commitment, evals-in-point-order)])],
    commitments in first-appearance order, sets in first-appearance order, points in ascending canonical order (BTreeSet)."""
    super_points = sorted({q.point for q in queries})
    order, by_key = [], {}
    for q in queries:
        k = key(q)
        if k not in by_key:
            by_key[k] = {"q": q, "points": {}}
            order.append(k)
        by_key[k]["points"].setdefault(q.point, q.eval)
    sets: List[Tuple[Tuple[int, ...], list]] = []
    for k in order:
        pts = tuple(sorted(by_key[k]["points"]))
        for s in sets:
            if s[0] == pts:
                s[1].append(k)
                break
        else:
            sets.append((pts, [k]))
    rotation_sets = [(pts, [(by_key[k]["q"], [by_key[k]["points"][p] for p in pts]) for k in ks]) for pts, ks in sets]
    return rotation_sets, super_points


def lagrange_interpolate(points: Sequence[int], evals: Sequence[int]) -> List[int]:
    """arithmetic::lagrange_interpolate: coefficients (low to high) of the unique poly of degree < len(points)."""
    n = len(points)
    coeffs = [0] * n
    for j in range(n):
        num = [1]                                   # prod_{m != j} (X - x_m)
        den = 1
        for m in range(n):
            if m == j:
                continue
            num = [(-points[m] * num[0]) % R_MOD] + [(num[i - 1] - points[m] * num[i]) % R_MOD for i in range(1, len(num))] + [num[-1]]
            den = den * (points[j] - points[m]) % R_MOD
        scale = evals[j] * pow(den, -1, R_MOD) % R_MOD
        for i, c in enumerate(num):
            coeffs[i] = (coeffs[i] + c * scale) % R_MOD
    return coeffs


def lagrange_basis(points: Sequence[int]) -> List[List[int]]:
    """coefficients of the Lagrange basis polynomials of a point set: lagrange_interpolate(points, evals) = sum_j evals[j] * basis[j].
    A rotation set interpolates every one of its commitments over the SAME points, so the products and the modular inverses are done once per set."""
    n = len(points)
    out = []
    for j in range(n):
        unit = [0] * n
        unit[j] = 1
        out.append(lagrange_interpolate(points, unit))
    return out


def interpolate_with_basis(basis: Sequence[Sequence[int]], evals: Sequence[int]) -> List[int]:
    n = len(basis)
    return [sum(evals[j] * basis[j][i] for j in range(n)) % R_MOD for i in range(n)]


def eval_poly_ints(coeffs: Sequence[int], x: int) -> int:
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % R_MOD
    return acc


def evaluate_vanishing_polynomial(roots: Sequence[int], z: int) -> int:
    acc = 1
    for r in roots:
        acc = acc * (z - r) % R_MOD
    return acc


class ProverSHPLONK:
    def __init__(self, params):
        self.params = params

    def create_proof(self, transcript, queries: Sequence[ProverQuery]) -> None:
        be, n = self.params.backend, self.params.n
        y = transcript.squeeze_challenge()
        rotation_sets, super_points = construct_intermediate_sets(queries)
        v = transcript.squeeze_challenge()

        pad = max(len(pts) for pts, _ in rotation_sets)              # largest rotation set: coefficients of the low-degree corrections / truncated tail
        rbuf = be.alloc(n * 32).zero()                                # carries the low-degree corrections (zero beyond a few coefficients)
        one = fr_mont(1)
        quotients, tmp = [], [be.alloc(n * 32), be.alloc(n * 32)]
        # Q_i(X) = sum_j y^j (P_ij(X) - R_ij(X)) / Z_i(X)
        low_degree = []
        for pts, commitments in rotation_sets:
            ypow, scal, polys, rsum = 1, [], [], [0] * len(pts)
            rs = []
            basis = lagrange_basis(pts)
            for q, evals in commitments:
                r = interpolate_with_basis(basis, evals)
                rs.append(r)
                for i, c in enumerate(r):
                    rsum[i] = (rsum[i] - ypow * c) % R_MOD
                polys.append(q.poly)
                scal.append(ypow)
                ypow = ypow * y % R_MOD
            low_degree.append(rs)
            rbuf.upload(fr_mont_array(rsum + [0] * (pad - len(rsum))))
            be.fr_lincomb_dev(polys + [rbuf], fr_mont_array(scal + [1]), n, tmp[0])
            cur, ln = 0, n
            for p in pts:                                             # div_by_vanishing: one synthetic division per point
                be.kate_division_dev(tmp[cur], ln, fr_mont(p), tmp[cur ^ 1])
                cur ^= 1
                ln -= 1
            qi = be.alloc(n * 32)
            be.fr_scale_dev(tmp[cur], one, qi, ln)
            qi.zero(ln * 32, (n - ln) * 32)                          # poly.resize(n, 0)
            quotients.append(qi)
        vp = [pow(v, i, R_MOD) for i in range(len(quotients))]
        h_x = be.alloc(n * 32)
        be.fr_lincomb_dev(quotients, fr_mont_array(vp), n, h_x)
        for qd in quotients:
            qd.free()
        transcript.write_point(g1_affine_ints(self.params.commit_columns("g", [h_x])[0]))
        u = transcript.squeeze_challenge()

        # L(X) = sum_i v^i z_i sum_j y^j (P_ij(X) - R_ij(u)) - Z_T(u) h(X), normalised by z_0, then divided by (X - u)
        z_diffs = []
        polys, scal, const = [], [], 0
        for i, (pts, commitments) in enumerate(rotation_sets):
            diffs = [p for p in super_points if p not in pts]
            z_i = evaluate_vanishing_polynomial(diffs, u)
            z_diffs.append(z_i)
            ypow = 1
            for (q, _), r in zip(commitments, low_degree[i]):
                w = vp[i] * z_i % R_MOD * ypow % R_MOD
                polys.append(q.poly)
                scal.append(w)
                const = (const - w * eval_poly_ints(r, u)) % R_MOD
                ypow = ypow * y % R_MOD
        zt = evaluate_vanishing_polynomial(super_points, u)
        z0_inv = pow(z_diffs[0], -1, R_MOD)
        polys.append(h_x)
        scal.append((-zt) % R_MOD)
        rbuf.upload(fr_mont_array([const * z0_inv % R_MOD] + [0] * (pad - 1)))
        be.fr_lincomb_dev(polys + [rbuf], fr_mont_array([s * z0_inv % R_MOD for s in scal] + [1]), n, tmp[0])
        be.kate_division_dev(tmp[0], n, fr_mont(u), tmp[1])
        tmp[1].zero((n - 1) * 32, 32)
        transcript.write_point(g1_affine_ints(self.params.commit_columns("g", [tmp[1]])[0]))
        for d in (h_x, rbuf, tmp[0], tmp[1]):
            d.free()
