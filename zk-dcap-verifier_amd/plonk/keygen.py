"""halo2_proofs::plonk::{keygen_vk, keygen_pk}, MI355X edition.

Mirrors (halo2_proofs 0.2.0 @ zkwebauthn c254c75, Cargo.lock:1314-1327) src/plonk/keygen.rs as the reference calls it at
circuits/src/sgx_dcap_verifier.rs:803,807: commit the fixed and permutation (sigma) columns (MSM),
bring them to coefficient form and to the extended coset (NTT), build l_0 / l_last / l_active_row, and compile the
Evaluator (`Evaluator::new(cs)`, src/plonk/evaluation.rs) — here into the ZKQ1 program the quotient kernel runs.
Everything a proof reuses stays resident in HBM inside the ProvingKey.

Deviation that cannot be closed without the pinned crate: `vk.transcript_repr` (halo2 hashes the Debug rendering of the
pinned verifying key) is replaced by a Blake2b digest of (k, column counts, commitments) — same role, different bytes.
"""
from __future__ import annotations

import hashlib
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from .. import evaluation as ev
from .._lib import Backend
from ..domain import EvaluationDomain
from ..fields import DELTA, R_MOD, fr_mont, fr_mont_array, g1_affine_ints
from ..kzg import ParamsKZG
from .circuit import Assembly, ConstraintSystem
from .expression import GraphBuilder


@dataclass
class VerifyingKey:
    k: int
    cs: ConstraintSystem
    fixed_commitments: list          # canonical affine (x, y) or None
    permutation_commitments: list
    transcript_repr: int = 0

    def hash_into(self, transcript) -> None:
        transcript.common_scalar(self.transcript_repr)


@dataclass
class ProvingKey:
    vk: VerifyingKey
    domain: EvaluationDomain
    backend: Backend
    fixed_values: list               # device, Lagrange
    fixed_polys: list                # device, coefficient form
    fixed_cosets: list               # device, extended coset
    sigma_values: list
    sigma_polys: list
    sigma_cosets: list
    l0: object
    l_last: object
    l_active_row: object
    evaluator: ev.Evaluator
    program: ev.Program
    lookup_compressors: list = field(default_factory=list)   # per lookup: (input Evaluator, table Evaluator) over the 2^k rows
    coset_parts: Optional[dict] = None   # params.by_cosets(): coset j -> {"fixed", "sigma", "l"}: the 2^k values of that coset only (this rank's cosets)
    borrowed: bool = False               # shared_with(): the columns belong to another ProvingKey of the process (same GPU); only the program handles are this key's
    owner: Optional["ProvingKey"] = None  # shared_with(): whose columns these are
    borrowers: int = 0                   # keys of other contexts that share this key's columns (an owner with borrowers refuses release())
    pieces_from_cosets: bool = False     # one GPU, cs_degree - 1 < 2^(extended_k - k): coset_parts holds cosets 0 .. cs_degree-2 and there are NO extended forms — the prover
                                         # evaluates h(X)'s numerator on those cosets only and takes the pieces from zk_cosets_to_pieces_dev

    @classmethod
    def shared_with(cls, other: "ProvingKey", backend: Backend) -> "ProvingKey":
        """The same proving key for another context on the same GPU (one context per host thread that proves concurrently): every column and every compiled
        program is SHARED with `other` — one proving key per process (≈2.3 GB at k = 19) instead of one per context, one keygen instead of N.  `other` must
        outlive the borrower's proofs; release() of a borrower returns only its program handles."""
        assert other.coset_parts is None or other.pieces_from_cosets, "sharded keys are per rank"
        ev_ = ev.Evaluator.shared(other.evaluator, backend)
        comps = [(ev.Evaluator.shared(a, backend), ev.Evaluator.shared(b, backend)) for a, b in other.lookup_compressors]
        root = other.owner or other
        root.borrowers += 1
        return cls(other.vk, EvaluationDomain(other.vk.cs.degree(), other.vk.k, backend=backend), backend, other.fixed_values, other.fixed_polys, other.fixed_cosets,
                   other.sigma_values, other.sigma_polys, other.sigma_cosets, other.l0, other.l_last, other.l_active_row, ev_, other.program, comps, other.coset_parts, True,
                   owner=root, pieces_from_cosets=other.pieces_from_cosets)

    def release(self):
        if not self.borrowed and self.borrowers:
            raise RuntimeError(f"ProvingKey.release: {self.borrowers} borrowed key(s) still share these columns (shared_with): release them first")
        self.evaluator.release()
        for a, b in self.lookup_compressors:
            a.release()
            b.release()
        if self.borrowed:
            if self.owner is not None:
                self.owner.borrowers -= 1
                self.owner = None
            return
        for grp in (self.fixed_values, self.fixed_polys, self.fixed_cosets, self.sigma_values, self.sigma_polys, self.sigma_cosets,
                    [d for d in (self.l0, self.l_last, self.l_active_row) if d is not None],
                    *[v for part in (self.coset_parts or {}).values() for v in part.values()]):
            for d in grp:
                d.free()


def compile_program(cs: ConstraintSystem, k: int, extended_k: int) -> ev.Program:
    """Evaluator::new(cs): custom gates folded with Horner in y; one graph per lookup ending in
    (theta-compressed input + beta) * (theta-compressed table + gamma)."""
    gb = GraphBuilder()
    parts = [gb.add_expression(g) for g in cs.gates]
    gb.add_calculation(ev.HORNER, ev.vs(ev.PREVIOUS), parts, ev.vs(ev.Y))
    lookups = []
    for lk in cs.lookups:
        lb = GraphBuilder()

        def lc(exprs, lb=lb):
            ps = [lb.add_expression(e) for e in exprs]
            return lb.add_calculation(ev.HORNER, ev.vs(ev.CONSTANT, 0), ps, ev.vs(ev.THETA))
        cin, ctab = lc(lk.input_expressions), lc(lk.table_expressions)
        right = lb.add_calculation(ev.ADD, ctab, ev.vs(ev.GAMMA))
        left = lb.add_calculation(ev.ADD, cin, ev.vs(ev.BETA))
        lb.add_calculation(ev.MUL, left, right)
        lookups.append(lb.graph)
    return ev.Program(k=k, extended_k=extended_k, n_fixed=cs.num_fixed_columns, n_advice=cs.num_advice_columns,
                      n_instance=cs.num_instance_columns, n_challenges=0, blinding_factors=cs.blinding_factors(), cs_degree=cs.degree(),
                      perm_columns=list(cs.permutation_columns), custom_gates=gb.graph, lookups=lookups)


def _compressor(cs: ConstraintSystem, k: int, exprs, be: Backend) -> ev.Evaluator:
    """lookup::Argument::commit_permuted's compress_expressions as a program over the Lagrange columns."""
    gb = GraphBuilder()
    ps = [gb.add_expression(e) for e in exprs]
    gb.add_calculation(ev.HORNER, ev.vs(ev.CONSTANT, 0), ps, ev.vs(ev.THETA))
    return ev.Evaluator(ev.expression_program(k, cs.num_fixed_columns, cs.num_advice_columns, cs.num_instance_columns, 0, gb.graph), backend=be)


def _as_mont(col, n) -> np.ndarray:
    if isinstance(col, np.ndarray) and col.dtype == np.uint64:
        a = np.ascontiguousarray(col).reshape(-1, 4)
    else:
        a = fr_mont_array(col)
    assert a.shape[0] == n, "column length must be 2^k"
    return a


def keygen(params: ParamsKZG, cs: ConstraintSystem, fixed_columns, assembly: Optional[Assembly] = None, piece_cosets: Optional[bool] = None) -> ProvingKey:
    """keygen_vk + keygen_pk.  fixed_columns: cs.num_fixed_columns columns of n = 2^k rows, either (n, 4) uint64
    Montgomery arrays or lists of canonical ints (selectors included, as halo2 hands them over after compression);
    assembly: the copy constraints (None = no equality-enabled columns).  piece_cosets (default on): on one GPU, when cs_degree - 1 is not a power of two, keep the
    key's columns on cosets 0 .. cs_degree-2 only (ProvingKey.pieces_from_cosets) — False builds the extended forms halo2's keygen_pk builds."""
    be, k, n = params.backend, params.k, params.n
    assert len(fixed_columns) == cs.num_fixed_columns
    dom = EvaluationDomain(cs.degree(), k, backend=be)
    ek = dom.extended_k
    bf = cs.blinding_factors()
    assert n > bf + 1 + 1, "circuit does not fit: not enough usable rows"

    fixed_values = [be.to_device(_as_mont(c, n)) for c in fixed_columns]
    # permutation columns: identity columns DELTA^j * omega^i = evaluations of DELTA^j * X (one NTT each), then the gather
    m = len(cs.permutation_columns)
    sigma_values = []
    if m:
        asm = assembly or Assembly(cs, k)
        ident = np.zeros((m, n, 4), dtype=np.uint64)
        bufs = []
        for j in range(m):
            col = np.zeros((n, 4), dtype=np.uint64)
            col[1] = fr_mont(pow(DELTA, j, R_MOD))
            d = be.to_device(col)
            be.coeff_to_lagrange_dev(d, k)
            bufs.append(d)
        for j, d in enumerate(bufs):
            ident[j] = d.download((n, 4))
            d.free()
        sig = asm.sigma_from_identity(ident)
        sigma_values = [be.to_device(np.ascontiguousarray(sig[j])) for j in range(m)]

    # commitments (keygen_vk): commit_lagrange of every fixed and sigma column — one batched MSM each
    fc = params.commit_columns("g_lagrange", fixed_values)
    pc = params.commit_columns("g_lagrange", sigma_values)
    fixed_commitments = [g1_affine_ints(r) for r in fc]
    permutation_commitments = [g1_affine_ints(r) for r in pc]
    h = hashlib.blake2b(digest_size=64, person=b"Halo2-Verify-Key")
    h.update(repr((k, cs.num_fixed_columns, cs.num_advice_columns, cs.num_instance_columns, len(cs.gates), len(cs.lookups),
                   cs.permutation_columns, cs.degree(), fixed_commitments, permutation_commitments)).encode())
    vk = VerifyingKey(k, cs, fixed_commitments, permutation_commitments, int.from_bytes(h.digest(), "little") % R_MOD)

    # keygen_pk: polys and extended cosets (when the quotient is sharded: only this rank's cosets, 2^k values each)
    by_cosets = params.by_cosets()
    my_cosets = params.my_cosets(1 << (ek - k)) if by_cosets else []
    n_pieces = cs.degree() - 1
    if piece_cosets is None:
        piece_cosets = True
    pieces_from_cosets = bool(piece_cosets) and not by_cosets and n_pieces < (1 << (ek - k)) and n_pieces <= 8
    if pieces_from_cosets:                                               # h(X) needs its numerator on cs_degree - 1 cosets only: keep those, build no extended form
        by_cosets, my_cosets = True, list(range(n_pieces))
    coset_parts = {j: {} for j in my_cosets} if by_cosets else None

    def to_poly_and_coset(values, name=None):
        polys = []
        for v in values:
            p = be.alloc(n * 32)
            be.fr_scale_dev(v, fr_mont(1), p, n)                         # device copy
            polys.append(p)
        if polys:
            be.lagrange_to_coeff_batch_dev(polys, k)
        if by_cosets:
            for j in my_cosets:
                outs = [be.alloc(n * 32) for _ in polys]
                if polys:
                    be.coeff_to_coset_batch_dev(polys, outs, k, ek, j)
                coset_parts[j][name] = outs
            return polys, []
        cosets = [be.alloc((1 << ek) * 32) for _ in polys]
        if polys:
            be.coeff_to_extended_batch_dev(polys, cosets, k, ek)
        return polys, cosets
    fixed_polys, fixed_cosets = to_poly_and_coset(fixed_values, "fixed")
    sigma_polys, sigma_cosets = to_poly_and_coset(sigma_values, "sigma")
    one = fr_mont(1)
    l0 = np.zeros((n, 4), dtype=np.uint64)
    l0[0] = one
    l_last = np.zeros((n, 4), dtype=np.uint64)
    l_last[n - bf - 1] = one
    l_act = np.zeros((n, 4), dtype=np.uint64)
    l_act[: n - bf - 1] = one                                            # 1 - (l_last + l_blind)
    lvals = [be.to_device(a) for a in (l0, l_last, l_act)]
    lpolys, lcosets = to_poly_and_coset(lvals, "l")
    for d in lvals + lpolys:
        d.free()
    if by_cosets:
        lcosets = [None, None, None]

    program = compile_program(cs, k, ek)
    evaluator = ev.Evaluator(program, backend=be)
    comps = [(_compressor(cs, k, lk.input_expressions, be), _compressor(cs, k, lk.table_expressions, be)) for lk in cs.lookups]
    return ProvingKey(vk, dom, be, fixed_values, fixed_polys, fixed_cosets, sigma_values, sigma_polys, sigma_cosets,
                      lcosets[0], lcosets[1], lcosets[2], evaluator, program, comps, coset_parts, False, pieces_from_cosets=pieces_from_cosets)
