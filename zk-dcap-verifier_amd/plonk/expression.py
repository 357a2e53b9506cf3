"""halo2_proofs::plonk::Expression and GraphEvaluator::add_expression, host side.

Mirrors (halo2_proofs 0.2.0 @ zkwebauthn c254c75, Cargo.lock:1314-1327) src/plonk/circuit.rs `Expression<F>` (the
polynomial tree a gate / lookup is written in — the sgx circuit builds them at
circuits/src/sgx_dcap_verifier.rs:86-137,139-238) and src/plonk/evaluation.rs `GraphEvaluator::add_expression`
(common-subexpression-eliminating compilation to the calculation list the quotient kernel runs).
Constants are canonical Python ints here and become Montgomery limbs when compiled.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Tuple

from .. import evaluation as ev
from ..fields import R_MOD, fr_mont


class Expression:
    def __add__(self, o): return Sum(self, _wrap(o))
    def __radd__(self, o): return Sum(_wrap(o), self)
    def __sub__(self, o): return Sum(self, Negated(_wrap(o)))
    def __rsub__(self, o): return Sum(_wrap(o), Negated(self))
    def __neg__(self): return Negated(self)

    def __mul__(self, o):
        if isinstance(o, int):
            return Scaled(self, o % R_MOD)
        return Product(self, o)

    def __rmul__(self, o):
        return self.__mul__(o)


def _wrap(x) -> "Expression":
    return Constant(x % R_MOD) if isinstance(x, int) else x


@dataclass(frozen=True, eq=True)
class Constant(Expression):
    value: int

@dataclass(frozen=True, eq=True)
class Fixed(Expression):
    column: int
    rotation: int = 0

@dataclass(frozen=True, eq=True)
class Advice(Expression):
    column: int
    rotation: int = 0

@dataclass(frozen=True, eq=True)
class Instance(Expression):
    column: int
    rotation: int = 0

@dataclass(frozen=True, eq=True)
class Negated(Expression):
    a: Expression

@dataclass(frozen=True, eq=True)
class Sum(Expression):
    a: Expression
    b: Expression

@dataclass(frozen=True, eq=True)
class Product(Expression):
    a: Expression
    b: Expression

@dataclass(frozen=True, eq=True)
class Scaled(Expression):
    a: Expression
    f: int


def degree(e: Expression) -> int:
    """Expression::degree."""
    if isinstance(e, Constant):
        return 0
    if isinstance(e, (Fixed, Advice, Instance)):
        return 1
    if isinstance(e, Negated):
        return degree(e.a)
    if isinstance(e, Sum):
        return max(degree(e.a), degree(e.b))
    if isinstance(e, Product):
        return degree(e.a) + degree(e.b)
    if isinstance(e, Scaled):
        return degree(e.a)
    raise TypeError(e)


def queries(e: Expression, out: dict) -> None:
    """collect (kind, column, rotation) queries in first-use order (cs.{advice,fixed,instance}_queries)."""
    if isinstance(e, Fixed):
        out.setdefault(("fixed", e.column, e.rotation), None)
    elif isinstance(e, Advice):
        out.setdefault(("advice", e.column, e.rotation), None)
    elif isinstance(e, Instance):
        out.setdefault(("instance", e.column, e.rotation), None)
    elif isinstance(e, (Negated, Scaled)):
        queries(e.a, out)
    elif isinstance(e, (Sum, Product)):
        queries(e.a, out)
        queries(e.b, out)


def evaluate(e: Expression, fixed, advice, instance) -> int:
    """Expression::evaluate over canonical ints; fixed/advice/instance: callables (column, rotation) -> int.
    (Host use: the verifier's gate check and small-k witness checks; never bulk data.)"""
    if isinstance(e, Constant):
        return e.value % R_MOD
    if isinstance(e, Fixed):
        return fixed(e.column, e.rotation)
    if isinstance(e, Advice):
        return advice(e.column, e.rotation)
    if isinstance(e, Instance):
        return instance(e.column, e.rotation)
    if isinstance(e, Negated):
        return (-evaluate(e.a, fixed, advice, instance)) % R_MOD
    if isinstance(e, Sum):
        return (evaluate(e.a, fixed, advice, instance) + evaluate(e.b, fixed, advice, instance)) % R_MOD
    if isinstance(e, Product):
        return evaluate(e.a, fixed, advice, instance) * evaluate(e.b, fixed, advice, instance) % R_MOD
    if isinstance(e, Scaled):
        return evaluate(e.a, fixed, advice, instance) * e.f % R_MOD
    raise TypeError(e)


class GraphBuilder:
    """GraphEvaluator with halo2's construction rules: constants start as [0, 1, 2]; `add_calculation` returns the
    existing intermediate when the same calculation was already emitted (CSE)."""

    def __init__(self):
        self.graph = ev.Graph()
        for c in (0, 1, 2):
            self.graph.constants.append(fr_mont(c))
        self._const_index = {0: 0, 1: 1, 2: 2}
        self._calc_index = {}

    def add_constant(self, value: int) -> Tuple[int, int, int]:
        value %= R_MOD
        if value not in self._const_index:
            self.graph.constants.append(fr_mont(value))
            self._const_index[value] = len(self.graph.constants) - 1
        return ev.vs(ev.CONSTANT, self._const_index[value])

    def add_rotation(self, rot: int) -> int:
        return self.graph.add_rotation(rot)

    def add_calculation(self, op: int, *operands) -> Tuple[int, int, int]:
        key = repr((op, operands))
        if key in self._calc_index:
            return ev.vs(ev.INTERMEDIATE, self._calc_index[key])
        target = self.graph.add_calculation(op, *operands)
        self._calc_index[key] = target[1]
        return target

    def add_expression(self, e: Expression) -> Tuple[int, int, int]:
        zero, one, two = ev.vs(ev.CONSTANT, 0), ev.vs(ev.CONSTANT, 1), ev.vs(ev.CONSTANT, 2)
        if isinstance(e, Constant):
            return self.add_constant(e.value)
        if isinstance(e, Fixed):
            return self.add_calculation(ev.STORE, ev.vs(ev.FIXED, e.column, self.add_rotation(e.rotation)))
        if isinstance(e, Advice):
            return self.add_calculation(ev.STORE, ev.vs(ev.ADVICE, e.column, self.add_rotation(e.rotation)))
        if isinstance(e, Instance):
            return self.add_calculation(ev.STORE, ev.vs(ev.INSTANCE, e.column, self.add_rotation(e.rotation)))
        if isinstance(e, Negated):
            if isinstance(e.a, Constant):
                return self.add_constant(-e.a.value)
            ra = self.add_expression(e.a)
            return ra if ra == zero else self.add_calculation(ev.NEGATE, ra)
        if isinstance(e, Sum):
            if isinstance(e.b, Negated):                       # a - b
                ra, rb = self.add_expression(e.a), self.add_expression(e.b.a)
                if ra == zero:
                    return self.add_calculation(ev.NEGATE, rb)
                if rb == zero:
                    return ra
                return self.add_calculation(ev.SUB, ra, rb)
            ra, rb = self.add_expression(e.a), self.add_expression(e.b)
            if ra == zero:
                return rb
            if rb == zero:
                return ra
            return self.add_calculation(ev.ADD, *sorted((ra, rb)))
        if isinstance(e, Product):
            ra, rb = self.add_expression(e.a), self.add_expression(e.b)
            if ra == zero or rb == zero:
                return zero
            if ra == one:
                return rb
            if rb == one:
                return ra
            if ra == two:
                return self.add_calculation(ev.DOUBLE, rb)
            if rb == two:
                return self.add_calculation(ev.DOUBLE, ra)
            if ra == rb:
                return self.add_calculation(ev.SQUARE, ra)
            return self.add_calculation(ev.MUL, *sorted((ra, rb)))
        if isinstance(e, Scaled):
            if e.f % R_MOD == 0:
                return zero
            cst = self.add_constant(e.f)
            ra = self.add_expression(e.a)
            if ra == zero:
                return ra
            if ra == one:
                return cst
            return self.add_calculation(ev.MUL, ra, cst)
        raise TypeError(e)
