"""halo2_proofs::dev::MockProver — the constraint check the reference runs before proving
(`MockProver::run(k, &circuit, vec![]).unwrap().assert_satisfied()`, circuits/src/sgx_dcap_verifier.rs:790-794).

Host-only and O(rows x constraints) in Python integers: meant for the small circuits of the tests, exactly as MockProver is
meant for debugging — it bypasses the commitment scheme (and therefore the GPU path) completely."""
from __future__ import annotations

from typing import List, Sequence

from ..fields import R_MOD, fr_int_array
from . import expression as ex
from .circuit import ADVICE, FIXED, INSTANCE, Assembly, ConstraintSystem


class VerifyFailure(Exception):
    pass


class MockProver:
    def __init__(self, k: int, cs: ConstraintSystem, fixed: Sequence, advice: Sequence, instances: Sequence[Sequence[int]], assembly: Assembly | None = None):
        self.k, self.n, self.cs, self.assembly = k, 1 << k, cs, assembly
        conv = lambda c: [int(v) % R_MOD for v in c] if not hasattr(c, "dtype") else fr_int_array(c)
        self.fixed = [conv(c) for c in fixed]
        self.advice = [conv(c) for c in advice]
        self.instance = [list(c) + [0] * (self.n - len(c)) for c in instances]

    @classmethod
    def run(cls, k, cs, fixed, advice, instances, assembly=None) -> "MockProver":
        return cls(k, cs, fixed, advice, instances, assembly)

    def verify(self) -> List[str]:
        cs, n = self.cs, self.n
        u = cs.usable_rows(self.k)
        failures: List[str] = []
        for row in range(u):
            fx = lambda c, r, row=row: self.fixed[c][(row + r) % n]
            ad = lambda c, r, row=row: self.advice[c][(row + r) % n]
            ins = lambda c, r, row=row: self.instance[c][(row + r) % n]
            for gi, g in enumerate(cs.gates):
                if ex.evaluate(g, fx, ad, ins) != 0:
                    failures.append(f"gate {gi} not satisfied on row {row}")
        for li, lk in enumerate(cs.lookups):
            rows = []
            for row in range(u):
                fx = lambda c, r, row=row: self.fixed[c][(row + r) % n]
                ad = lambda c, r, row=row: self.advice[c][(row + r) % n]
                ins = lambda c, r, row=row: self.instance[c][(row + r) % n]
                rows.append((tuple(ex.evaluate(e, fx, ad, ins) for e in lk.input_expressions),
                             tuple(ex.evaluate(e, fx, ad, ins) for e in lk.table_expressions)))
            table = {t for _, t in rows}
            for row, (inp, _) in enumerate(rows):
                if inp not in table:
                    failures.append(f"lookup {li}: input of row {row} is not in the table")
        if self.assembly is not None:
            cols = {ADVICE: self.advice, FIXED: self.fixed, INSTANCE: self.instance}
            for j, (t, i) in enumerate(self.assembly.columns):
                for row in range(n):
                    cj, rj = int(self.assembly.map_c[j, row]), int(self.assembly.map_r[j, row])
                    if (cj, rj) != (j, row):
                        t2, i2 = self.assembly.columns[cj]
                        if cols[t][i][row] != cols[t2][i2][rj]:
                            failures.append(f"copy constraint violated: column {j} row {row} != column {cj} row {rj}")
        return failures

    def assert_satisfied(self) -> None:
        f = self.verify()
        if f:
            raise VerifyFailure("; ".join(f[:5]) + (f" (+{len(f) - 5} more)" if len(f) > 5 else ""))
