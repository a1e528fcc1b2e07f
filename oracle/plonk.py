"""TEST INFRASTRUCTURE ONLY -- constraint-system pieces of the CPU oracle beyond CQ lookups.

Restates, with Python ints, the parts of the reference a general (gates + fixed/instance
columns + copy constraints) circuit adds to the CQ-only path:

* `Expression` (plonk/circuit.rs:780-1100) as nested tuples, its degree and evaluation
  (`evaluate`, plonk/evaluation.rs:776-818; `get_rotation_idx` :37-39);
* `permutation::keygen::Assembly` (plonk/permutation/keygen.rs:14-113) and `build_pk` (:151-208);
* `permutation::Argument::commit` (plonk/permutation/prover.rs:47-198);
* the legacy (plookup-style) lookup argument: `permute_expression_pair`, `commit_permuted`, `commit_product`
  (plonk/lookup/prover.rs:57-300, 400-502) and its terms of `evaluate_h` (plonk/evaluation.rs:461-531);
* the permutation terms of `evaluate_h` (plonk/evaluation.rs:367-459).

Parity status: no reference test pins any of these numerically (the permutation argument is only
exercised end to end, `halo2_proofs/tests/plonk_api.rs`, with OsRng) -> "parity unpinned" beyond
the verifier-side identities checked in `tests/test_oracle_plonk.py`.
"""
from __future__ import annotations

from .bn254 import FR_DELTA, R_MOD, fr_random
from .poly import batch_invert

P = R_MOD

ADVICE, FIXED, INSTANCE = "advice", "fixed", "instance"


# ---- Expression -----------------------------------------------------------------------------------
def const(v):
    return ("const", v % P)


def adv(col, rot=0):
    return (ADVICE, col, rot)


def fix(col, rot=0):
    return (FIXED, col, rot)


def inst(col, rot=0):
    return (INSTANCE, col, rot)


def chal(idx):
    """`Expression::Challenge` (circuit.rs:793-794): the idx-th user challenge of the constraint system."""
    return ("challenge", idx)


def neg(a):
    return ("neg", a)


def add(a, b):
    return ("add", a, b)


def sub(a, b):
    """`a - b` is `a + (-b)` (circuit.rs:1105-1110)."""
    return ("add", a, ("neg", b))


def mul(a, b):
    return ("mul", a, b)


def scale(a, v):
    return ("scale", a, v % P)


def expr_degree(e) -> int:
    """circuit.rs:1040-1056."""
    t = e[0]
    if t in ("const", "challenge"):
        return 0
    if t in (ADVICE, FIXED, INSTANCE):
        return 1
    if t in ("neg", "scale"):
        return expr_degree(e[1])
    if t == "add":
        return max(expr_degree(e[1]), expr_degree(e[2]))
    if t == "mul":
        return expr_degree(e[1]) + expr_degree(e[2])
    raise ValueError(t)


def expr_queries(e, out):
    """Column queries in traversal order (the order `meta.query_*` calls register them in)."""
    t = e[0]
    if t in (ADVICE, FIXED, INSTANCE):
        if (t, e[1], e[2]) not in out:
            out.append((t, e[1], e[2]))
    elif t in ("neg", "scale"):
        expr_queries(e[1], out)
    elif t in ("add", "mul"):
        expr_queries(e[1], out)
        expr_queries(e[2], out)


def expr_eval(e, get, challenges=()):
    """`Expression::evaluate` (circuit.rs:880-960); `get(kind, col, rot)` resolves a query, `challenges` the
    user challenges squeezed so far."""
    t = e[0]
    if t == "const":
        return e[1]
    if t == "challenge":
        return challenges[e[1]]
    if t in (ADVICE, FIXED, INSTANCE):
        return get(t, e[1], e[2])
    if t == "neg":
        return (-expr_eval(e[1], get, challenges)) % P
    if t == "add":
        return (expr_eval(e[1], get, challenges) + expr_eval(e[2], get, challenges)) % P
    if t == "mul":
        return expr_eval(e[1], get, challenges) * expr_eval(e[2], get, challenges) % P
    if t == "scale":
        return expr_eval(e[1], get, challenges) * e[2] % P
    raise ValueError(t)


def rotation_idx(idx: int, rot: int, rot_scale: int, size: int) -> int:
    """evaluation.rs:37-39."""
    return (idx + rot * rot_scale) % size


# ---- permutation keygen ---------------------------------------------------------------------------
class Assembly:
    """permutation/keygen.rs:14-113: cycles of equal cells kept as a mapping + union-by-size."""

    def __init__(self, n: int, columns):
        self.columns = list(columns)
        m = len(self.columns)
        self.mapping = [[(i, j) for j in range(n)] for i in range(m)]
        self.aux = [[(i, j) for j in range(n)] for i in range(m)]
        self.sizes = [[1] * n for _ in range(m)]

    def copy(self, left_column, left_row, right_column, right_row):
        lc = self.columns.index(left_column)
        rc = self.columns.index(right_column)
        if left_row >= len(self.mapping[lc]) or right_row >= len(self.mapping[rc]):
            raise IndexError("BoundsFailure")
        left_cycle = self.aux[lc][left_row]
        right_cycle = self.aux[rc][right_row]
        if left_cycle == right_cycle:
            return
        if self.sizes[left_cycle[0]][left_cycle[1]] < self.sizes[right_cycle[0]][right_cycle[1]]:
            left_cycle, right_cycle = right_cycle, left_cycle
        self.sizes[left_cycle[0]][left_cycle[1]] += self.sizes[right_cycle[0]][right_cycle[1]]
        i = right_cycle
        while True:
            self.aux[i[0]][i[1]] = left_cycle
            i = self.mapping[i[0]][i[1]]
            if i == right_cycle:
                break
        self.mapping[lc][left_row], self.mapping[rc][right_row] = self.mapping[rc][right_row], self.mapping[lc][left_row]


def build_permutation_pk(dom, mapping):
    """permutation/keygen.rs:151-208 -> (permutations, polys, cosets)."""
    n = dom.n
    omega_powers = [1] * n
    for i in range(1, n):
        omega_powers[i] = omega_powers[i - 1] * dom.omega % P
    deltaomega = []
    cur = 1
    for _ in mapping:
        deltaomega.append([o * cur % P for o in omega_powers])
        cur = cur * FR_DELTA % P
    perms, polys, cosets = [], [], []
    for col in mapping:
        lag = [deltaomega[pi][pj] for (pi, pj) in col]
        perms.append(lag)
        poly = dom.lagrange_to_coeff(lag)
        polys.append(poly)
        cosets.append(dom.coeff_to_extended(poly))
    return perms, polys, cosets


# ---- permutation prover ---------------------------------------------------------------------------
def permutation_commit(dom, cs_degree, bf, columns, column_values, permutations, beta, gamma, rng):
    """permutation/prover.rs:47-198 up to the Lagrange values of every z.

    columns: the argument's columns; column_values(col) -> the n Lagrange values of that column.
    Returns [(z_lagrange, blind)] per set, drawing `bf` blinding rows + 1 blind per set from rng."""
    n = dom.n
    assert cs_degree >= 3
    chunk_len = cs_degree - 2
    deltaomega = 1
    last_z = 1
    sets = []
    for c0 in range(0, len(columns), chunk_len):
        cols = columns[c0 : c0 + chunk_len]
        perms = permutations[c0 : c0 + chunk_len]
        modified = [1] * n
        for col, perm in zip(cols, perms):  # :106-121
            vals = column_values(col)
            modified = [m * ((beta * s + gamma + v) % P) % P for m, s, v in zip(modified, perm, vals)]
        modified = batch_invert(modified)  # :124
        for col in cols:  # :128-147
            vals = column_values(col)
            dw = deltaomega
            out = []
            for m, v in zip(modified, vals):
                out.append(m * ((dw * beta + gamma + v) % P) % P)
                dw = dw * dom.omega % P
            modified = out
            deltaomega = deltaomega * FR_DELTA % P
        z = [last_z]  # :160-166
        for row in range(1, n):
            z.append(z[row - 1] * modified[row - 1] % P)
        for r in range(n - bf, n):  # :169-171
            z[r] = fr_random(rng)
        last_z = z[n - (bf + 1)]  # :173
        blind = fr_random(rng)  # :175
        sets.append((z, blind))
    return sets


def permutation_h_terms(dom, cs_degree, bf, columns, column_coset, perm_cosets, z_cosets, l0, l_last, l_active, beta,
                        gamma, y, h):
    """evaluation.rs:367-459: folds the permutation constraints into `h` (extended coset), in place."""
    if not z_cosets:
        return h
    size = dom.extended_len
    rot_scale = 1 << (dom.extended_k - dom.k)
    chunk_len = cs_degree - 2
    from .bn254 import FR_ZETA

    delta_start = beta * FR_ZETA % P
    last_rotation = -(bf + 1)
    chunks = [columns[i : i + chunk_len] for i in range(0, len(columns), chunk_len)]
    beta_term = 1
    first, last = z_cosets[0], z_cosets[-1]
    out = list(h)
    for idx in range(size):
        r_next = rotation_idx(idx, 1, rot_scale, size)
        r_last = rotation_idx(idx, last_rotation, rot_scale, size)
        v = out[idx]
        v = (v * y + (1 - first[idx]) * l0[idx]) % P
        v = (v * y + (last[idx] * last[idx] - last[idx]) * l_last[idx]) % P
        for si in range(1, len(z_cosets)):
            v = (v * y + (z_cosets[si][idx] - z_cosets[si - 1][r_last]) * l0[idx]) % P
        current_delta = delta_start * beta_term % P
        ci = 0
        for si, cols in enumerate(chunks):
            left = z_cosets[si][r_next]
            for col in cols:
                left = left * ((column_coset(col)[idx] + beta * perm_cosets[ci][idx] + gamma) % P) % P
                ci += 1
            right = z_cosets[si][idx]
            for col in cols:
                right = right * ((column_coset(col)[idx] + current_delta + gamma) % P) % P
                current_delta = current_delta * FR_DELTA % P
            v = (v * y + (left - right) * l_active[idx]) % P
        out[idx] = v
        beta_term = beta_term * dom.extended_omega % P
    return out


# ---- legacy lookup argument (plonk/lookup/prover.rs) ----------------------------------------------------------
def permute_expression_pair(n, bf, input_expression, table_expression, rng):
    """lookup/prover.rs:400-502.  `sort()` and the BTreeMap order field elements by canonical value
    (derive/field.rs `Ord`: big-endian comparison of `to_repr`)."""
    usable = n - (bf + 1)
    permuted_input = sorted(input_expression[:usable])
    leftover = {}
    for v in table_expression[:usable]:
        leftover[v] = leftover.get(v, 0) + 1
    permuted_table = [0] * usable
    repeated = []
    for row, v in enumerate(permuted_input):
        if row == 0 or v != permuted_input[row - 1]:
            permuted_table[row] = v
            if leftover.get(v, 0) > 0:
                leftover[v] -= 1
            else:
                raise ValueError("ConstraintSystemFailure: lookup input not in table")
        else:
            repeated.append(row)
    for v in sorted(leftover):
        for _ in range(leftover[v]):
            permuted_table[repeated.pop()] = v
    assert not repeated
    permuted_input += [fr_random(rng) for _ in range(bf + 1)]
    permuted_table += [fr_random(rng) for _ in range(bf + 1)]
    return permuted_input, permuted_table


def lookup_product(n, bf, compressed_input, compressed_table, permuted_input, permuted_table, beta, gamma, rng):
    """lookup/prover.rs:163-300: Lagrange values of the grand product z (without its blind)."""
    prod = [((beta + a) % P) * ((gamma + s_) % P) % P for a, s_ in zip(permuted_input, permuted_table)]
    prod = batch_invert(prod)
    prod = [p_ * ((ci + beta) % P) % P * ((ct + gamma) % P) % P for p_, ci, ct in zip(prod, compressed_input, compressed_table)]
    z = [1]
    for v in prod:
        z.append(z[-1] * v % P)
    z = z[: n - bf]
    z += [fr_random(rng) for _ in range(bf)]
    assert len(z) == n
    return z


def lookup_h_terms(dom, z_coset, a_coset, s_coset, table_value, l0, l_last, l_active, beta, gamma, y, h):
    """evaluation.rs:461-531; `table_value[idx]` = (compressed input + beta)(compressed table + gamma) on the coset."""
    size = dom.extended_len
    rot_scale = 1 << (dom.extended_k - dom.k)
    out = list(h)
    for idx in range(size):
        r_next = rotation_idx(idx, 1, rot_scale, size)
        r_prev = rotation_idx(idx, -1, rot_scale, size)
        a_minus_s = (a_coset[idx] - s_coset[idx]) % P
        v = out[idx]
        v = (v * y + (1 - z_coset[idx]) * l0[idx]) % P
        v = (v * y + (z_coset[idx] * z_coset[idx] - z_coset[idx]) * l_last[idx]) % P
        v = (v * y + (z_coset[r_next] * ((a_coset[idx] + beta) % P) % P * ((s_coset[idx] + gamma) % P)
                      - z_coset[idx] * table_value[idx]) % P * l_active[idx]) % P
        v = (v * y + a_minus_s * l0[idx]) % P
        v = (v * y + a_minus_s * ((a_coset[idx] - a_coset[r_prev]) % P) % P * l_active[idx]) % P
        out[idx] = v
    return out
