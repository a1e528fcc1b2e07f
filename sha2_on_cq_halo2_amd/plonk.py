"""Host-side mirror of the reference's circuit-description API for the general-PLONK part of the
proving path: `Expression` (plonk/circuit.rs:780-1100), `ConstraintSystem` (circuit.rs:1380-2060:
column allocation, `query_*`, `create_gate`, `enable_equality`, `lookup_static`, `degree`,
`blinding_factors`) and `permutation::keygen::Assembly` (plonk/permutation/keygen.rs:14-113).

It only DESCRIBES a circuit and lowers it to the plain arrays of `cq_plonk` (include/cq_halo2.h);
all field / curve arithmetic of the proof happens in libcq_halo2.so.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

FR_MODULUS = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001

GATE_CONST, GATE_ADVICE, GATE_FIXED, GATE_INSTANCE, GATE_NEG, GATE_ADD, GATE_MUL, GATE_SCALE, GATE_CHALLENGE = range(9)
COL_ADVICE, COL_FIXED, COL_INSTANCE = 0, 1, 2
_QUERY_OP = {COL_ADVICE: GATE_ADVICE, COL_FIXED: GATE_FIXED, COL_INSTANCE: GATE_INSTANCE}


class Column:
    """`Column<Any>` (circuit.rs:24-160)."""

    __slots__ = ("kind", "index")

    def __init__(self, kind: int, index: int):
        self.kind, self.index = kind, index

    def __eq__(self, o):
        return isinstance(o, Column) and (self.kind, self.index) == (o.kind, o.index)

    def __hash__(self):
        return hash((self.kind, self.index))

    def __repr__(self):
        return f"Column({('advice', 'fixed', 'instance')[self.kind]}, {self.index})"


class Expression:
    """`Expression<F>` (circuit.rs:780-806): Constant, Fixed/Advice/Instance queries, Negated, Sum,
    Product, Scaled.  (`Selector` never reaches the prover: selectors are compressed into fixed columns,
    keygen.rs:318-326.)"""

    __slots__ = ("op", "a", "b")

    def __init__(self, op, a=None, b=None):
        self.op, self.a, self.b = op, a, b

    @staticmethod
    def constant(v: int) -> "Expression":
        return Expression("const", v % FR_MODULUS)

    @staticmethod
    def query(column: Column, rotation: int = 0) -> "Expression":
        return Expression("query", column, rotation)

    @staticmethod
    def challenge(index: int) -> "Expression":
        """`Expression::Challenge` (circuit.rs:793-794)."""
        return Expression("challenge", index)

    def __neg__(self):
        return Expression("neg", self)

    def __add__(self, o):
        return Expression("add", self, _expr(o))

    def __radd__(self, o):
        return Expression("add", _expr(o), self)

    def __sub__(self, o):  # `a - b` is `a + (-b)` (circuit.rs:1105-1110)
        return Expression("add", self, Expression("neg", _expr(o)))

    def __mul__(self, o):
        if isinstance(o, int):  # Mul<F> -> Scaled (circuit.rs:1122-1127)
            return Expression("scale", self, o % FR_MODULUS)
        return Expression("mul", self, o)

    def __rmul__(self, o):
        return self.__mul__(o)

    def degree(self) -> int:
        """circuit.rs:1040-1056."""
        if self.op in ("const", "challenge"):
            return 0
        if self.op == "query":
            return 1
        if self.op in ("neg", "scale"):
            return self.a.degree()
        if self.op == "add":
            return max(self.a.degree(), self.b.degree())
        return self.a.degree() + self.b.degree()

    def compile(self, constants: list) -> list:
        """Postfix program of u32 words for the GPU gate interpreter (include/cq_halo2.h, CQ_GATE_*)."""

        def const_index(v):
            if v not in constants:
                constants.append(v)
            return constants.index(v)

        out = []

        def walk(e):
            if e.op == "const":
                out.append(GATE_CONST | const_index(e.a) << 8)
            elif e.op == "challenge":
                out.append(GATE_CHALLENGE | e.a << 8)
            elif e.op == "query":
                out.append(_QUERY_OP[e.a.kind] | e.a.index << 8)
                out.append(e.b & 0xFFFFFFFF)
            elif e.op == "neg":
                walk(e.a)
                out.append(GATE_NEG)
            elif e.op == "scale":
                walk(e.a)
                out.append(GATE_SCALE | const_index(e.b) << 8)
            else:
                walk(e.a)
                walk(e.b)
                out.append(GATE_ADD if e.op == "add" else GATE_MUL)

        walk(self)
        return out


def _expr(o) -> Expression:
    return o if isinstance(o, Expression) else Expression.constant(int(o))


class ConstraintSystem:
    """`ConstraintSystem<F>` (circuit.rs:1380-1470) as the prover reads it from `vk.cs`."""

    def __init__(self):
        self.num_advice_columns = 0
        self.num_fixed_columns = 0
        self.num_instance_columns = 0
        self.advice_queries = []   # (column index, rotation), registration order
        self.fixed_queries = []
        self.instance_queries = []
        self.gates = []            # flat list of gate polynomials (Gate::polynomials, cs.gates order)
        self.permutation_columns = []
        self.static_lookups = []   # [[(advice column index, table)]]
        self._lookup_exprs = []    # the input expression of every (lookup, table column) pair, flattened
        self.lookups = []          # legacy lookups: [([input expressions], [table expressions])] (plonk/lookup.rs:9-36)
        self.advice_column_phase = []  # phase of every advice column (circuit.rs `advice_column_phase`)
        self.challenge_phase = []      # phase after which each user challenge becomes available

    # -- columns (circuit.rs:1903-1960)
    def advice_column(self, phase: int = 0) -> Column:
        """`advice_column` / `advice_column_in(phase)` (circuit.rs:1903-1935)."""
        self.num_advice_columns += 1
        self.advice_column_phase.append(phase)
        return Column(COL_ADVICE, self.num_advice_columns - 1)

    def challenge_usable_after(self, phase: int) -> Expression:
        """circuit.rs:1962-1977: a challenge squeezed once the commitments of `phase` are in the transcript."""
        self.challenge_phase.append(phase)
        return Expression.challenge(len(self.challenge_phase) - 1)

    def fixed_column(self) -> Column:
        self.num_fixed_columns += 1
        return Column(COL_FIXED, self.num_fixed_columns - 1)

    def instance_column(self) -> Column:
        self.num_instance_columns += 1
        return Column(COL_INSTANCE, self.num_instance_columns - 1)

    # -- queries (circuit.rs:1604-1681): first-seen order
    def _queries(self, kind):
        return (self.advice_queries, self.fixed_queries, self.instance_queries)[kind]

    def query_any(self, column: Column, rotation: int = 0) -> Expression:
        qs = self._queries(column.kind)
        if (column.index, rotation) not in qs:
            qs.append((column.index, rotation))
        return Expression.query(column, rotation)

    query_advice = query_fixed = query_instance = query_any

    def enable_equality(self, column: Column):
        """circuit.rs:1523-1527."""
        self.query_any(column, 0)
        if column not in self.permutation_columns:
            self.permutation_columns.append(column)

    def create_gate(self, name: str, polys):
        """circuit.rs:1683-1724 (constraints already multiplied by their selector expressions)."""
        polys = list(polys)
        assert polys, "Gates must contain at least one constraint."
        self.gates.extend(polys)

    def lookup(self, name: str, table_map):
        """`lookup_any` (circuit.rs:1556-1577): [(input Expression, table Expression)] -- the legacy plookup-style
        argument (plonk/lookup.rs); `lookup` with TableColumns is the same with fixed-column table expressions."""
        ins, tabs = [i for i, _ in table_map], [t for _, t in table_map]
        assert ins, "a lookup needs at least one (input, table) pair"
        self.lookups.append((ins, tabs))
        return len(self.lookups) - 1

    def lookup_static(self, name: str, table_map):
        """circuit.rs:1579-1602: [(input Expression (or a Column, queried at Rotation::cur()), StaticTable)]."""
        row = []
        for expr, table in table_map:
            if isinstance(expr, Column):
                expr = self.query_any(expr, 0)
            plain = expr.op == "query" and expr.a.kind == COL_ADVICE and expr.b == 0
            row.append((expr.a.index if plain else 0, table))
            self._lookup_exprs.append(expr)
        self.static_lookups.append(row)

    @property
    def static_lookup_inputs(self):
        """None when every input is `advice[col] @ Rotation::cur()` (the CQ-only fast path), else all inputs
        as expressions."""
        if all(e.op == "query" and e.a.kind == COL_ADVICE and e.b == 0 for e in self._lookup_exprs):
            return None
        return self._lookup_exprs

    def degree(self) -> int:
        """circuit.rs:1979-2018: permutation argument 3, static lookups 3 (static_lookup.rs:181-190), gates."""
        d = 3
        for g in self.gates:
            d = max(d, g.degree())
        for e in self._lookup_exprs:
            d = max(d, 2 + e.degree())
        for ins, tabs in self.lookups:  # lookup.rs:37-52
            d = max(d, 4, 2 + max([1] + [e.degree() for e in ins]) + max([1] + [e.degree() for e in tabs]))
        return d

    def blinding_factors(self) -> int:
        """circuit.rs:2022-2047."""
        per = [0] * self.num_advice_columns
        for col, _ in self.advice_queries:
            per[col] += 1
        return max(3, max(per) if per else 1) + 2

    def minimum_rows(self) -> int:
        """circuit.rs:2051-2059."""
        return self.blinding_factors() + 3


class PermutationAssembly:
    """`permutation::keygen::Assembly` (permutation/keygen.rs:14-113) over libcq_halo2's host bookkeeping."""

    def __init__(self, lib, n: int, columns):
        self.lib, self.n, self.columns = lib, n, list(columns)
        m = len(self.columns)
        self.mapping = np.zeros((m, n, 2), dtype=np.uint32)
        self._aux = np.zeros((m, n, 2), dtype=np.uint32)
        self._sizes = np.zeros((m, n), dtype=np.uint32)
        p = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint32))
        lib.cq_permutation_assembly_init(m, n, p(self.mapping), p(self._aux), p(self._sizes))

    def copy(self, left_column: Column, left_row: int, right_column: Column, right_row: int):
        try:
            lc, rc = self.columns.index(left_column), self.columns.index(right_column)
        except ValueError:
            raise ValueError("Error::ColumnNotInPermutation") from None
        p = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint32))
        rc_ = self.lib.cq_permutation_assembly_copy(len(self.columns), self.n, p(self.mapping), p(self._aux), p(self._sizes),
                                                    lc, left_row, rc, right_row)
        if rc_ != 0:
            raise IndexError("Error::BoundsFailure")
