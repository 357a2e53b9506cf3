"""halo2_proofs::plonk::{ConstraintSystem, permutation::keygen::Assembly}, host side.

Mirrors (halo2_proofs 0.2.0 @ zkwebauthn c254c75, Cargo.lock:1314-1327) src/plonk/circuit.rs `ConstraintSystem<F>` —
what `Circuit::configure` fills in (the sgx circuit: circuits/src/sgx_dcap_verifier.rs:139-238) — restricted to what
the prover's hot path reads: column counts, gate polynomials, lookup arguments, the permutation's column list, the
query lists, `degree()` and `blinding_factors()`.  Selectors are ordinary fixed columns
here (halo2 compresses them into fixed columns before keygen, so the prover never sees a selector either).
The circuit source itself (chips, regions, layouter, witness synthesis) is host code the north star leaves
unchanged; this mirror only describes its *output*.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Sequence, Tuple

from ..fields import DELTA, R_MOD, omega
from . import expression as ex

ADVICE, FIXED, INSTANCE = 0, 1, 2      # column_type numbering of the ZKQ1 blob


@dataclass
class LookupArgument:
    """plonk::lookup::Argument { input_expressions, table_expressions }"""
    input_expressions: List[ex.Expression]
    table_expressions: List[ex.Expression]

    def required_degree(self) -> int:
        din = max([1] + [ex.degree(e) for e in self.input_expressions])
        dt = max([1] + [ex.degree(e) for e in self.table_expressions])
        return max(4, 2 + din + dt)


@dataclass
class ConstraintSystem:
    num_fixed_columns: int = 0
    num_advice_columns: int = 0
    num_instance_columns: int = 0
    gates: List[ex.Expression] = field(default_factory=list)            # every polynomial of every gate, in order
    lookups: List[LookupArgument] = field(default_factory=list)
    permutation_columns: List[Tuple[int, int]] = field(default_factory=list)   # (column_type, index) — enable_equality order
    minimum_degree: int = 1

    # -- construction (ConstraintSystem::{create_gate, lookup, enable_equality}) -------------------------------------
    # halo2 records a column query the moment `meta.query_*` runs, i.e. in CALL order across create_gate / lookup / enable_equality
    # (enable_equality queries the column at Rotation::cur() itself): the query log below is kept the same way, so the order of the advice /
    # fixed evaluations in the proof follows the order of the configure() calls, as in the reference.
    _query_log: Dict[Tuple[str, int, int], None] = field(default_factory=dict, repr=False, compare=False)

    def create_gate(self, *polys: ex.Expression) -> None:
        self.gates.extend(polys)
        for g in polys:
            ex.queries(g, self._query_log)

    def lookup(self, pairs: Sequence[Tuple[ex.Expression, ex.Expression]]) -> int:
        self.lookups.append(LookupArgument([p[0] for p in pairs], [p[1] for p in pairs]))
        for inp, tab in pairs:
            ex.queries(inp, self._query_log)
            ex.queries(tab, self._query_log)
        return len(self.lookups) - 1

    def enable_equality(self, column_type: int, index: int) -> None:
        # permutation::Argument::add_column + query_any_index(column, Rotation::cur())
        self._query_log.setdefault(({ADVICE: "advice", FIXED: "fixed", INSTANCE: "instance"}[column_type], index, 0), None)
        if (column_type, index) not in self.permutation_columns:
            self.permutation_columns.append((column_type, index))

    # -- derived quantities -------------------------------------------------------------------------------------------
    def _queries(self) -> Dict[Tuple[str, int, int], None]:
        return self._query_log

    def advice_queries(self) -> List[Tuple[int, int]]:
        return [(c, r) for (k, c, r) in self._queries() if k == "advice"]

    def fixed_queries(self) -> List[Tuple[int, int]]:
        return [(c, r) for (k, c, r) in self._queries() if k == "fixed"]

    def instance_queries(self) -> List[Tuple[int, int]]:
        return [(c, r) for (k, c, r) in self._queries() if k == "instance"]

    def degree(self) -> int:
        d = 3                                                           # permutation::Argument::required_degree(): 3 whether or not a column is equality-enabled
        for lk in self.lookups:
            d = max(d, lk.required_degree())
        for g in self.gates:
            d = max(d, ex.degree(g))
        return max(d, self.minimum_degree)

    def blinding_factors(self) -> int:
        per_col = [0] * max(1, self.num_advice_columns)
        for c, _ in self.advice_queries():
            per_col[c] += 1
        return max(3, max(per_col)) + 2

    def permutation_chunk_len(self) -> int:
        return self.degree() - 2

    def usable_rows(self, k: int) -> int:
        return (1 << k) - (self.blinding_factors() + 1)


class Assembly:
    """plonk::permutation::keygen::Assembly: union of copy-constraint cycles over the equality-enabled columns.
    Cell successors are kept as two int32 planes (column, row) so a k = 19 circuit costs a few hundred MB, not GBs."""

    def __init__(self, cs: ConstraintSystem, k: int):
        import numpy as np
        self.columns = list(cs.permutation_columns)
        self.n = 1 << k
        m = len(self.columns)
        rows = np.arange(self.n, dtype=np.int32)
        self.map_c = np.repeat(np.arange(m, dtype=np.int32)[:, None], self.n, axis=1)
        self.map_r = np.repeat(rows[None, :], m, axis=0)
        self.aux_c, self.aux_r = self.map_c.copy(), self.map_r.copy()
        self.sizes = np.ones((m, self.n), dtype=np.int32)
        self.copies = None          # set to [] to have every copy constraint logged as ((type, index, row), (type, index, row)) — small circuits only

    def copy(self, left: Tuple[int, int, int], right: Tuple[int, int, int]) -> None:
        """left/right = (column_type, index, row).  Mirrors Assembly::copy (cycle merge by swapping successors)."""
        if self.copies is not None:
            self.copies.append((tuple(int(v) for v in left), tuple(int(v) for v in right)))
        lc = self.columns.index((left[0], left[1]))
        rc = self.columns.index((right[0], right[1]))
        lr, rr = left[2], right[2]
        lcyc = (int(self.aux_c[lc, lr]), int(self.aux_r[lc, lr]))
        rcyc = (int(self.aux_c[rc, rr]), int(self.aux_r[rc, rr]))
        if lcyc == rcyc:
            return
        if self.sizes[lcyc] < self.sizes[rcyc]:
            lcyc, rcyc = rcyc, lcyc
            lc, lr, rc, rr = rc, rr, lc, lr
        self.sizes[lcyc] += self.sizes[rcyc]
        i, j = rcyc
        while True:                                                     # relabel the smaller cycle
            self.aux_c[i, j], self.aux_r[i, j] = lcyc
            i, j = int(self.map_c[i, j]), int(self.map_r[i, j])
            if (i, j) == rcyc:
                break
        a = (self.map_c[lc, lr], self.map_r[lc, lr])
        self.map_c[lc, lr], self.map_r[lc, lr] = self.map_c[rc, rr], self.map_r[rc, rr]
        self.map_c[rc, rr], self.map_r[rc, rr] = a

    def copy_rows(self, left: Tuple[int, int], right: Tuple[int, int], rows) -> None:
        """Vectorised `copy((left, r), (right, r)) for r in rows` for cells that are still unconstrained (singleton cycles) —
        the bulk case of a region that copies a whole column range; falls back to `copy` row by row otherwise."""
        import numpy as np
        lc, rc = self.columns.index(left), self.columns.index(right)
        rows = np.asarray(rows, dtype=np.int64)
        def singleton(c):        # `sizes` is kept for cycle representatives only: a merged non-representative cell still reads 1 there
            return (self.aux_c[c, rows] == c) & (self.aux_r[c, rows] == rows) & (self.sizes[c, rows] == 1)
        if self.copies is not None or lc == rc or not singleton(lc).all() or not singleton(rc).all() or np.unique(rows).size != rows.size:
            for r in rows.tolist():
                self.copy((left[0], left[1], r), (right[0], right[1], r))
            return
        self.map_c[lc, rows], self.map_r[lc, rows] = rc, rows
        self.map_c[rc, rows], self.map_r[rc, rows] = lc, rows
        self.aux_c[rc, rows], self.aux_r[rc, rows] = lc, rows
        self.sizes[lc, rows] = 2

    def sigma_columns(self, k: int) -> List[List[int]]:
        """build_pk's permutation polynomials in Lagrange form as canonical ints (small k / tests):
        sigma_j[i] = DELTA^(j') * omega^(i') for (j', i') = mapping[j][i].  keygen_pk uses the array form below."""
        w = omega(k)
        wp = [1] * self.n
        for i in range(1, self.n):
            wp[i] = wp[i - 1] * w % R_MOD
        dp = [pow(DELTA, j, R_MOD) for j in range(len(self.columns))]
        return [[dp[int(self.map_c[j, i])] * wp[int(self.map_r[j, i])] % R_MOD for i in range(self.n)] for j in range(len(self.columns))]

    def sigma_from_identity(self, ident):
        """ident: (m, n, 4) uint64 — column j = the Montgomery limbs of DELTA^j * omega^i (built on the GPU as the NTT of
        DELTA^j * X).  Returns the (m, n, 4) sigma columns: a pure gather through the successor planes."""
        return ident[self.map_c, self.map_r]
