"""halo2_proofs::plonk::evaluation, MI355X edition.

Mirrors (halo2_proofs 0.2.0 @ zkwebauthn c254c75) src/plonk/evaluation.rs: GraphEvaluator (constants,
rotations, calculations over ValueSource) and Evaluator::evaluate_h, reached
from the reference through create_proof (circuits/src/sgx_dcap_verifier.rs:814-822).

A proving key's compiled evaluator is serialised once into a "ZKQ1" blob (little-endian u32 words;
layout documented in INTEGRATION.md) — `Program.to_blob()` below is the Python twin of the
serialiser the Rust shim carries.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import List, Tuple

import numpy as np

from ._lib import Backend, default_backend

# ValueSource kinds / Calculation opcodes (numbering of the ZKQ1 format)
CONSTANT, INTERMEDIATE, FIXED, ADVICE, INSTANCE, CHALLENGE, BETA, GAMMA, THETA, Y, PREVIOUS = range(11)
ADD, SUB, MUL, SQUARE, DOUBLE, NEGATE, HORNER, STORE = range(8)
MAGIC = 0x31514B5A  # 'ZKQ1'

VS = Tuple[int, int, int]  # (kind, a, b)


def vs(kind: int, a: int = 0, b: int = 0) -> VS:
    return (kind, a, b)


@dataclass
class Graph:
    """GraphEvaluator: constants are Fr Montgomery limbs (4 x u64 each)."""
    constants: List[np.ndarray] = field(default_factory=list)
    rotations: List[int] = field(default_factory=list)
    calculations: list = field(default_factory=list)   # (op, target, operands...)
    num_intermediates: int = 0

    def add_constant(self, limbs) -> VS:
        limbs = np.asarray(limbs, dtype=np.uint64).reshape(4)
        for i, c in enumerate(self.constants):
            if (c == limbs).all():
                return vs(CONSTANT, i)
        self.constants.append(limbs)
        return vs(CONSTANT, len(self.constants) - 1)

    def add_rotation(self, rot: int) -> int:
        if rot in self.rotations:
            return self.rotations.index(rot)
        self.rotations.append(rot)
        return len(self.rotations) - 1

    def add_calculation(self, op: int, *operands) -> VS:
        target = self.num_intermediates
        self.num_intermediates += 1
        self.calculations.append((op, target, operands))
        return vs(INTERMEDIATE, target)

    def words(self) -> List[int]:
        w = [len(self.constants)]
        for c in self.constants:
            for limb in c:
                w += [int(limb) & 0xFFFFFFFF, int(limb) >> 32]
        w.append(len(self.rotations))
        w += [r & 0xFFFFFFFF for r in self.rotations]
        w.append(self.num_intermediates)
        w.append(len(self.calculations))
        for op, target, ops in self.calculations:
            w += [op, target]
            if op in (ADD, SUB, MUL):
                a, b = ops
                w += list(a) + list(b)
            elif op == HORNER:
                start, parts, factor = ops
                w += list(start) + list(factor) + [len(parts)]
                for p in parts:
                    w += list(p)
            else:
                (a,) = ops
                w += list(a)
        return w


@dataclass
class Program:
    k: int
    extended_k: int
    n_fixed: int
    n_advice: int
    n_instance: int
    n_challenges: int
    blinding_factors: int
    cs_degree: int
    perm_columns: List[Tuple[int, int]]      # (0 advice | 1 fixed | 2 instance, index)
    custom_gates: Graph
    lookups: List[Graph]

    def to_blob(self) -> bytes:
        w = [MAGIC, self.k, self.extended_k, self.n_fixed, self.n_advice, self.n_instance, self.n_challenges,
             self.blinding_factors, self.cs_degree, len(self.perm_columns)]
        for t, i in self.perm_columns:
            w += [t, i]
        w.append(len(self.lookups))
        w += self.custom_gates.words()
        for g in self.lookups:
            w += g.words()
        return struct.pack("<%dI" % len(w), *w)


class Evaluator:
    """Holds the program resident on the GPU; evaluate_h runs it on device-resident cosets."""

    def __init__(self, program: Program, backend: Backend | None = None):
        self.backend = backend or default_backend()
        self.program = program
        self.handle = self.backend.quotient_program_load(program.to_blob())

    @classmethod
    def shared(cls, other: "Evaluator", backend: Backend) -> "Evaluator":
        """the same compiled program for another context on the same GPU (zk_quotient_program_share): nothing is compiled or uploaded again"""
        self = cls.__new__(cls)
        self.backend, self.program = backend, other.program
        self.handle = backend.quotient_program_share(other.backend, other.handle)
        return self

    def evaluate_h(self, *, fixed, advice, instance, l0, l_last, l_active_row, perm_cosets, perm_products,
                   lookup_product, lookup_input, lookup_table, challenges, beta, gamma, theta, y, out, coset: int | None = None, rows: tuple | None = None,
                   part: int = 0, low_cosets: int = 0):
        """All columns are device buffers holding 2^extended_k Fr values (extended cosets);
        `out` receives the numerator of h(X) on the extended coset (before divide_by_vanishing_poly).  With `coset = j` every column holds the
        2^k values of coset j only (coeff_to_coset) and `out` that coset's numerator values — the unit a multi-GPU prover shards by;
        `rows = (lo, count)` restricts the run to that aligned power-of-two slice of the coset's rows (ranks that share a coset).
        `part` = 1 / 2: only the identities of degree above / up to 3 of a program that carries the degree split (Backend.quotient_program_split; include/zkmi355.h);
        part 2 with `low_cosets` evaluates them on the rows of cosets 0 .. low_cosets-1 of extended-layout columns, `out` coset-major."""
        self.backend.quotient_run_dev(self.handle, fixed=fixed, advice=advice, instance=instance, l0=l0, l_last=l_last,
                                      l_active_row=l_active_row, perm_cosets=perm_cosets, perm_products=perm_products,
                                      lookup_product=lookup_product, lookup_input=lookup_input, lookup_table=lookup_table,
                                      challenges=challenges, beta=beta, gamma=gamma, theta=theta, y=y, out=out, coset=coset, rows=rows, part=part, low_cosets=low_cosets)

    # -- the shape of halo2's own call: polynomials in, polynomial out -----------------------------------
    def load_pk(self, fixed, sigma, l0, l_last, l_active_row, extended: bool = False) -> int:
        """Upload what keygen_pk keeps for the evaluator (host arrays): coefficient form, or the extended
        cosets pk already stores (extended=True).  Returns a pk handle for evaluate_h_polys."""
        self.pk = self.backend.pk_load(self.handle, fixed, sigma, l0, l_last, l_active_row, form=1 if extended else 0)
        return self.pk

    def evaluate_h_polys(self, *, advice, instance, perm_products, lookup_product, lookup_input, lookup_table, challenges, beta, gamma,
                         theta, y, finish: bool = True):
        """Evaluator::evaluate_h on host coefficient-form polynomials; finish=True also applies
        divide_by_vanishing_poly + extended_to_coeff and returns the (cs_degree - 1) * n coefficients of h(X)."""
        p = self.program
        rows = (1 << p.k) * (p.cs_degree - 1) if finish else 1 << p.extended_k
        return self.backend.evaluate_h(self.pk, advice=advice, instance=instance, perm_products=perm_products, lookup_product=lookup_product,
                                       lookup_input=lookup_input, lookup_table=lookup_table, challenges=challenges, beta=beta, gamma=gamma,
                                       theta=theta, y=y, out_rows=rows, finish=finish)

    def release(self):
        if getattr(self, "pk", 0):
            self.backend.pk_release(self.pk)
            self.pk = 0
        if self.handle:
            self.backend.quotient_program_release(self.handle)
            self.handle = 0


def expression_program(k: int, n_fixed: int, n_advice: int, n_instance: int, n_challenges: int, graph: Graph) -> Program:
    """A Program that evaluates ONE expression graph over the n = 2^k rows of the (non-extended) domain — rotations wrap
    modulo n, exactly like `Expression::evaluate` over Lagrange columns.  With the graph `Horner(e_0, [e_1 .. e_m-1], Theta)`
    this is lookup::Argument::commit_permuted's `compress_expressions` (theta-compression of the input / table expressions,
    halo2_proofs src/plonk/lookup/prover.rs; SURVEY 8f n4, first half).  Run it with Evaluator.evaluate_h on Lagrange columns:
    l0 / l_last / l_active_row are unused and may be any column."""
    return Program(k=k, extended_k=k, n_fixed=n_fixed, n_advice=n_advice, n_instance=n_instance, n_challenges=n_challenges,
                   blinding_factors=0, cs_degree=3, perm_columns=[], custom_gates=graph, lookups=[])
