"""TEST INFRASTRUCTURE ONLY -- CPU oracle restating `create_proof`.

Follows `halo2_proofs/src/plonk/prover.rs:51-779` for one circuit with advice, fixed and instance
columns, custom gates, the permutation argument and static (CQ) lookups -- everything the
reference prover does except legacy (plookup-style) lookups, multi-phase challenges and
SHPLONK.  The reference's one CQ end-to-end test (`halo2_proofs/tests/my_test.rs`) is the
advice + static-lookup subset.  Sub-arguments: `plonk/static_lookup/prover.rs`,
`plonk/permutation/prover.rs` (oracle/plonk.py), `plonk/vanishing/prover.rs`,
`plonk/evaluation.rs:285-551`, `poly/kzg/multiopen/gwc/prover.rs`, `transcript.rs`.

Parity status: the arithmetic underneath is pinned by the reference's KATs
(`oracle/bn254.py`); the proof byte stream itself is "parity unpinned" against
the Rust prover (no golden proof exists in the reference: its test draws from
OsRng, my_test.rs:221) and is instead checked by the verifier-side identities
in `tests/test_oracle_prover.py`.
"""
from __future__ import annotations

import hashlib
from dataclasses import dataclass, field

from .bn254 import (
    JAC_ID,
    R_MOD,
    batch_to_affine,
    fr_random,
    from_bytes_wide,
    g1_to_bytes,
    inv_mod,
    jac_add,
    jac_mul,
    jac_to_affine,
    to_jac,
    to_repr,
)
from .plonk import (ADVICE, FIXED, INSTANCE, build_permutation_pk, expr_degree, expr_eval, expr_queries, lookup_h_terms,
                    lookup_product, permutation_commit, permutation_h_terms, permute_expression_pair, rotation_idx)
from .poly import EvaluationDomain, best_multiexp, eval_polynomial, kate_division
from .shplonk import shplonk_prove

P = R_MOD


class Blake2bWrite:
    """transcript.rs:170-241 with Challenge255 (:287-315)."""

    def __init__(self):
        self.state = hashlib.blake2b(digest_size=64, person=b"Halo2-Transcript")
        self.proof = bytearray()

    def common_scalar(self, s: int):
        self.state.update(b"\x02")
        self.state.update(to_repr(s))

    def common_point(self, A):
        if A is None:
            raise ValueError("cannot write points at infinity to the transcript")
        self.state.update(b"\x01")
        self.state.update(to_repr(A[0]))
        self.state.update(to_repr(A[1]))

    def write_point(self, A):
        self.common_point(A)
        self.proof += g1_to_bytes(A)

    def write_scalar(self, s: int):
        self.common_scalar(s)
        self.proof += to_repr(s)

    def squeeze_challenge_scalar(self) -> int:
        self.state.update(b"\x00")
        return from_bytes_wide(self.state.copy().digest())


@dataclass
class CqCircuit:
    """Shape descriptor standing in for `ConstraintSystem` (plonk/circuit.rs).  `lookups[l]` =
    list of (input, table id), one per table column, as `lookup_static` registers them
    (plonk/circuit.rs:1579-1602); an input is an advice column index (`advice[col]@Rotation::cur()`)
    or any oracle/plonk.py expression.

    General-PLONK part (all empty for a CQ-only circuit): `gates` = the gate polynomials in
    `cs.gates` order (oracle/plonk.py expressions; selectors already folded into fixed columns as
    in `vk.cs`), `perm_columns` = `cs.permutation.columns` as (kind, index), fixed / instance
    column counts.  `queries` optionally pins the (advice, fixed, instance) query lists; by default
    they are registered as a `configure` that calls enable_equality, then create_gate, then
    lookup_static would (first-seen order, circuit.rs:1619-1681)."""

    k: int
    num_advice: int
    lookups: list
    num_fixed: int = 0
    num_instance: int = 0
    gates: list = field(default_factory=list)
    perm_columns: list = field(default_factory=list)
    queries: dict = None
    # legacy (plookup-style) lookups, `cs.lookups`: [(input expressions, table expressions)] (plonk/lookup.rs:9-36)
    plookups: list = field(default_factory=list)
    # phases (circuit.rs `advice_column_phase` / `challenge_phase`): advice column c is committed in phase
    # advice_phases[c] (default: all 0); user challenge i is squeezed after the commitments of challenge_phases[i]
    advice_phases: list = None
    challenge_phases: list = field(default_factory=list)

    def phase_of(self, col):
        return self.advice_phases[col] if self.advice_phases else 0

    def num_phases(self):
        return 1 + max([self.phase_of(c) for c in range(self.num_advice)] + list(self.challenge_phases) + [0])

    def _all_queries(self):
        if self.queries is not None:
            return self.queries
        q = []
        for kind, idx in self.perm_columns:  # enable_equality -> query_any_index(col, cur) (circuit.rs:1523-1527)
            if (kind, idx, 0) not in q:
                q.append((kind, idx, 0))
        for ins, tabs in self.plookups:
            for e in list(ins) + list(tabs):
                expr_queries(e, q)
        for g in self.gates:
            expr_queries(g, q)
        for lk in self.lookups:
            for col, _ in lk:
                expr_queries(_lookup_input(col), q)
        return {kind: [(c, r) for (t, c, r) in q if t == kind] for kind in (ADVICE, FIXED, INSTANCE)}

    def advice_queries(self):
        """plonk/circuit.rs:1619-1633: queries in first-seen order."""
        return self._all_queries()[ADVICE]

    def fixed_queries(self):
        return self._all_queries()[FIXED]

    def instance_queries(self):
        return self._all_queries()[INSTANCE]

    def blinding_factors(self) -> int:
        """plonk/circuit.rs:2022-2047."""
        per_col = [0] * self.num_advice
        for col, _ in self.advice_queries():
            per_col[col] += 1
        factors = max(per_col) if per_col else 1
        return max(3, factors) + 2

    def degree(self) -> int:
        """plonk/circuit.rs:1979-2018: permutation 3 (permutation.rs:40-75), static lookups 3
        (static_lookup.rs:181-190), gates by their polynomials."""
        d = 3
        for g in self.gates:
            d = max(d, expr_degree(g))
        for ins, tabs in self.plookups:  # lookup.rs:37-52: max(4, 2 + input degree + table degree)
            d = max(d, 4, 2 + max([1] + [expr_degree(e) for e in ins]) + max([1] + [expr_degree(e) for e in tabs]))
        for lk in self.lookups:  # static_lookup.rs:181-190: max(3, 2 + input degree)
            for col, _ in lk:
                d = max(d, 2 + expr_degree(_lookup_input(col)))
        return d


def _lookup_input(col):
    """A static lookup input: an advice column index (`advice[col] @ Rotation::cur()`) or any expression."""
    return (ADVICE, col, 0) if isinstance(col, int) else col


@dataclass
class ProvingKey:
    """The slice of `ProvingKey` (plonk.rs:291-308) this path reads."""

    circuit: CqCircuit
    domain: EvaluationDomain
    l_active_row: list  # extended coset, keygen.rs:366-373
    tables: dict  # id -> StaticTableValues
    table_cfg: object  # TableSRS (g1_lagrange, g_lagrange_opening_at_0), static_lookup.rs:47-66
    b0_g1_bound: list
    vk_repr: int  # opaque transcript_repr (plonk.rs:221-232)
    l0: list = None  # keygen.rs:340-345
    l_last: list = None  # :357-363
    fixed_values: list = field(default_factory=list)  # :320-326
    fixed_polys: list = field(default_factory=list)  # :328-331
    fixed_cosets: list = field(default_factory=list)  # :333-336
    permutations: list = field(default_factory=list)  # permutation/keygen.rs:151-208
    perm_polys: list = field(default_factory=list)
    perm_cosets: list = field(default_factory=list)


def keygen_pk(circuit: CqCircuit, tables: dict, table_cfg, b0_g1_bound, vk_repr: int, fixed=(), perm_mapping=None) -> ProvingKey:
    """plonk/keygen.rs:278-397.  `fixed`: the assigned fixed columns (n values each, after
    `compress_selectors`); `perm_mapping`: `Assembly.mapping` after the circuit's copy constraints
    (identity if None)."""
    dom = EvaluationDomain(circuit.degree(), circuit.k)
    n = dom.n
    bf = circuit.blinding_factors()
    l_blind = [0] * n
    for i in range(n - bf, n):
        l_blind[i] = 1
    l_last = [0] * n
    l_last[n - bf - 1] = 1
    l0 = [0] * n
    l0[0] = 1
    lb = dom.coeff_to_extended(dom.lagrange_to_coeff(l_blind))
    ll = dom.coeff_to_extended(dom.lagrange_to_coeff(l_last))
    l0e = dom.coeff_to_extended(dom.lagrange_to_coeff(l0))
    l_active = [(1 - (a + b)) % P for a, b in zip(ll, lb)]
    assert len(b0_g1_bound) == n - 1 or not circuit.lookups
    assert len(fixed) == circuit.num_fixed
    fixed_values = [[v % P for v in col] + [0] * (n - len(col)) for col in fixed]
    fixed_polys = [dom.lagrange_to_coeff(c) for c in fixed_values]
    fixed_cosets = [dom.coeff_to_extended(c) for c in fixed_polys]
    if perm_mapping is None:
        perm_mapping = [[(i, j) for j in range(n)] for i in range(len(circuit.perm_columns))]
    assert len(perm_mapping) == len(circuit.perm_columns)
    perms, ppolys, pcosets = build_permutation_pk(dom, perm_mapping)
    return ProvingKey(circuit, dom, l_active, tables, table_cfg, b0_g1_bound, vk_repr, l0e, ll, fixed_values,
                      fixed_polys, fixed_cosets, perms, ppolys, pcosets)


@dataclass
class ProofTrace:
    """Intermediate values exposed so tests can compare stage by stage."""

    proof: bytes = b""
    challenges: dict = field(default_factory=dict)
    points: dict = field(default_factory=dict)
    scalars: dict = field(default_factory=dict)
    polys: dict = field(default_factory=dict)


def create_proof(params, pk: ProvingKey, advice_usable, rng, msm=None, instances=(), opener="gwc") -> ProofTrace:
    """plonk/prover.rs:51-779 for a single circuit (ProverGWC: QUERY_INSTANCE = false).

    `advice_usable[c]` = the assigned values of advice column c on rows 0..u (shorter
    lists are zero-padded: unassigned cells are zero, prover.rs:424); for a column of a later phase
    it may be a callable `f(challenges) -> values`, evaluated when that phase starts (the witness of
    a later phase depends on the challenges of the earlier ones, prover.rs:436-463).
    `msm(coeffs, bases) -> Jacobian` lets tests swap in a faster multiexp.
    """
    msm = msm or best_multiexp
    cs = pk.circuit
    dom = pk.domain
    n = dom.n
    bf = cs.blinding_factors()
    u = n - (bf + 1)
    tr = Blake2bWrite()
    out = ProofTrace()

    def commit_affine(jac):
        return jac_to_affine(jac)

    # prover.rs:85
    tr.common_scalar(pk.vk_repr)

    # ---- instance columns (prover.rs:100-131); absorbed as scalars (:305-312) ---
    assert len(instances) == cs.num_instance, "InvalidInstances"
    instance_values = []
    for vals in instances:
        assert len(vals) <= u, "InstanceTooLarge"
        instance_values.append([v % P for v in vals] + [0] * (n - len(vals)))
    instance_polys = [dom.lagrange_to_coeff(v) for v in instance_values]
    for vals in instances:
        for v in vals:
            tr.common_scalar(v % P)

    # ---- advice, phase by phase (prover.rs:299-391, 436-463) ----------------------
    advice = [None] * cs.num_advice
    adv_cm = [None] * cs.num_advice
    challenges = [None] * len(cs.challenge_phases)
    for phase in range(cs.num_phases()):
        cols = [c for c in range(cs.num_advice) if cs.phase_of(c) == phase]
        for c in cols:
            vals = advice_usable[c](list(challenges)) if callable(advice_usable[c]) else advice_usable[c]
            col = list(vals) + [0] * (u - len(vals))
            assert len(col) == u, "advice assigned outside usable rows"
            advice[c] = [v % P for v in col] + [0] * (n - u)
        for c in cols:  # :346-350
            for r in range(u, n):
                advice[c][r] = fr_random(rng)
        for c in cols:  # :352-355 (blinds drawn, unused by KZG)
            fr_random(rng)
        cms = batch_to_affine([msm(advice[c], params.g_lagrange) for c in cols])
        for c, A in zip(cols, cms):
            tr.write_point(A)
            adv_cm[c] = A
        for i, ph in enumerate(cs.challenge_phases):  # :383-389
            if ph == phase:
                challenges[i] = tr.squeeze_challenge_scalar()
    out.points["advice"] = adv_cm
    out.challenges["user"] = list(challenges)

    theta = tr.squeeze_challenge_scalar()  # :472
    out.challenges["theta"] = theta

    # ---- legacy lookups: commit_permuted (lookup/prover.rs:57-160) -------------
    cols_now = {ADVICE: advice, FIXED: pk.fixed_values, INSTANCE: instance_values}

    def compress_lagrange(exprs):  # `evaluate(expr, n, 1, ..)` folded with theta (:98-117)
        acc = [0] * n
        for e in exprs:
            vals = [expr_eval(e, lambda kind, c_, rot, row=row: cols_now[kind][c_][rotation_idx(row, rot, 1, n)], challenges) for row in range(n)]
            acc = [(a * theta + b) % P for a, b in zip(acc, vals)]
        return acc

    permuted = []
    for ins, tabs in cs.plookups:
        ci, ct = compress_lagrange(ins), compress_lagrange(tabs)
        pi, pt = permute_expression_pair(n, bf, ci, ct, rng)
        fr_random(rng)  # permuted_input_blind (:133-137)
        pi_cm = commit_affine(msm(pi, params.g_lagrange))
        fr_random(rng)  # permuted_table_blind
        pt_cm = commit_affine(msm(pt, params.g_lagrange))
        tr.write_point(pi_cm)
        tr.write_point(pt_cm)
        permuted.append((ci, ct, pi, pt, dom.lagrange_to_coeff(pi), dom.lagrange_to_coeff(pt)))

    # ---- CQ round 1 (static_lookup/prover.rs:51-183) -----------------------
    committed = []
    for li, lk in enumerate(cs.lookups):
        tables = [pk.tables[tid] for _, tid in lk]
        assert all(t.size == tables[0].size for t in tables)
        cols_now = {ADVICE: advice, FIXED: pk.fixed_values, INSTANCE: instance_values}
        exprs = []
        for col, _ in lk:  # `evaluate(expression, n, 1, ...)` (:91-107, evaluation.rs:776-818)
            e = _lookup_input(col)
            exprs.append([expr_eval(e, lambda kind, c, rot, row=row: cols_now[kind][c][rotation_idx(row, rot, 1, n)], challenges)
                          for row in range(n)])
        f = [0] * n
        for e in exprs:  # :108-116
            f = [(a * theta + b) % P for a, b in zip(f, e)]
        m_sparse = {}
        tivm = [dict() for _ in tables]
        for row in range(u):  # :132-161
            idx = None
            for ti, (e, t) in enumerate(zip(exprs, tables)):
                fi = e[row]
                if fi not in t.value_index_mapping:
                    raise KeyError(f"{fi} not in table")
                index = t.value_index_mapping[fi]
                tivm[ti][index] = fi
                if idx is not None and idx != index:
                    raise ValueError("Vector lookup must be on the same table row")
                idx = index
            m_sparse[idx] = m_sparse.get(idx, 0) + 1
        f_cm = commit_affine(msm(f, params.g_lagrange))
        m_cm = JAC_ID
        for index in sorted(m_sparse):  # :167-170
            m_cm = jac_add(jac_mul(to_jac(pk.table_cfg.g1_lagrange[index]), m_sparse[index]), m_cm)
        m_cm = commit_affine(m_cm)
        tr.write_point(f_cm)
        tr.write_point(m_cm)
        out.points[f"f_cm{li}"] = f_cm
        out.points[f"m_cm{li}"] = m_cm
        committed.append((f, m_sparse, tables, tivm))

    beta = tr.squeeze_challenge_scalar()  # :529
    gamma = tr.squeeze_challenge_scalar()  # :532
    out.challenges["beta"] = beta
    out.challenges["gamma"] = gamma

    # ---- permutation commit (permutation/prover.rs:47-198) ------------------
    def column_values(col):
        kind, idx = col
        return {ADVICE: advice, FIXED: pk.fixed_values, INSTANCE: instance_values}[kind][idx]

    perm_sets = []
    if cs.perm_columns:
        for z, _blind in permutation_commit(dom, cs.degree(), bf, cs.perm_columns, column_values, pk.permutations, beta, gamma, rng):
            z_cm = commit_affine(msm(z, params.g_lagrange))
            tr.write_point(z_cm)
            perm_sets.append(dom.lagrange_to_coeff(z))
            out.points.setdefault("perm_z", []).append(z_cm)

    # ---- legacy lookups: commit_product (lookup/prover.rs:163-300) --------------
    plk = []
    for (ci, ct, pi, pt, pi_poly, pt_poly) in permuted:
        z = lookup_product(n, bf, ci, ct, pi, pt, beta, gamma, rng)
        fr_random(rng)  # product_blind (:283)
        tr.write_point(commit_affine(msm(z, params.g_lagrange)))
        plk.append((dom.lagrange_to_coeff(z), pi_poly, pt_poly))

    # ---- CQ round 2 (static_lookup/prover.rs:187-342) ----------------------
    logd = []
    size_n = pk.table_cfg.size
    for li, (f, m_sparse, tables, tivm) in enumerate(committed):
        a_cm = JAC_ID
        qa_cm = JAC_ID
        a0_cm = JAC_ID
        f_set = set(f)
        for index in sorted(m_sparse):  # :245-257
            tv = 0
            tq = JAC_ID
            for ti, t in enumerate(tables):  # compress_tables :224-240
                tv = (tv * theta + tivm[ti][index]) % P
                tq = jac_add(jac_mul(tq, theta), to_jac(t.qs[index]))
            a_i = m_sparse[index] * inv_mod((tv + beta) % P, P) % P
            assert tv in f_set  # :250
            a_cm = jac_add(jac_mul(to_jac(pk.table_cfg.g1_lagrange[index]), a_i), a_cm)
            qa_cm = jac_add(jac_mul(tq, a_i), qa_cm)
            a0_cm = jac_add(jac_mul(to_jac(pk.table_cfg.g_lagrange_opening_at_0[index]), a_i), a0_cm)
        bs = [inv_mod((fi + beta) % P, P) for fi in f[:u]]  # :261-266
        beta_inv = inv_mod(beta, P)
        bs += [beta_inv] * (bf + 1)  # :268-269
        b_poly = dom.ifft(bs, dom.omega_inv, dom.k, dom.ifft_divisor)  # :271-276
        b0 = b_poly[1:]  # :279
        p_cm = commit_affine(msm(b0, pk.b0_g1_bound))  # :299
        b0 = b0 + [0]  # :301
        for Pt in (a_cm, qa_cm, a0_cm):  # :306-308
            tr.write_point(commit_affine(Pt))
        b0_cm = commit_affine(msm(b0, params.g))  # :310
        tr.write_point(b0_cm)  # :312
        tr.write_point(p_cm)  # :313
        out.points[f"a_cm{li}"] = commit_affine(a_cm)
        out.points[f"qa_cm{li}"] = commit_affine(qa_cm)
        out.points[f"a0_cm{li}"] = commit_affine(a0_cm)
        out.points[f"b0_cm{li}"] = b0_cm
        out.points[f"p_cm{li}"] = p_cm
        b_at_zero = b_poly[0]  # :318
        a_at_zero = (b_at_zero * n - (bf + 1) * beta_inv) % P * inv_mod(size_n % P, P) % P  # :319-324
        f_coeff = dom.lagrange_to_coeff(f)  # :326-334
        logd.append((b_poly, b0, f_coeff, a_at_zero))
        out.polys[f"b{li}"] = b_poly
        out.polys[f"f{li}"] = f_coeff

    # ---- vanishing commit (vanishing/prover.rs:37-65) ----------------------
    random_poly = [fr_random(rng) for _ in range(n)]
    fr_random(rng)  # random_blind
    rnd_cm = commit_affine(msm(random_poly, params.g))
    tr.write_point(rnd_cm)
    out.points["random"] = rnd_cm

    y = tr.squeeze_challenge_scalar()  # prover.rs:584
    out.challenges["y"] = y

    advice_polys = [dom.lagrange_to_coeff(a) for a in advice]  # :587-603

    # ---- evaluate_h (evaluation.rs:285-551): gates, permutation, CQ terms ---
    ext = dom.extended_len
    rot_scale = 1 << (dom.extended_k - dom.k)
    h = [0] * ext
    advice_cosets = instance_cosets = None
    if cs.gates or cs.perm_columns or cs.plookups:
        advice_cosets = [dom.coeff_to_extended(p_) for p_ in advice_polys]  # :317-335
        instance_cosets = [dom.coeff_to_extended(p_) for p_ in instance_polys]
        src = {ADVICE: advice_cosets, FIXED: pk.fixed_cosets, INSTANCE: instance_cosets}
        for idx in range(ext):  # custom gates, Horner in y (:226-235, :348-365)
            get = lambda kind, col, rot: src[kind][col][rotation_idx(idx, rot, rot_scale, ext)]
            v = h[idx]
            for g in cs.gates:
                v = (v * y + expr_eval(g, get, challenges)) % P
            h[idx] = v
        z_cosets = [dom.coeff_to_extended(z) for z in perm_sets]
        h = permutation_h_terms(dom, cs.degree(), bf, cs.perm_columns, lambda col: src[col[0]][col[1]], pk.perm_cosets,
                                z_cosets, pk.l0, pk.l_last, pk.l_active_row, beta, gamma, y, h)
    for (ins, tabs), (z_poly, pi_poly, pt_poly) in zip(cs.plookups, plk):  # legacy lookups (:461-531)
        def compress_coset(exprs):
            acc = [0] * ext
            for e in exprs:
                vals = [expr_eval(e, lambda kind, c_, rot, idx=idx: src[kind][c_][rotation_idx(idx, rot, rot_scale, ext)], challenges) for idx in range(ext)]
                acc = [(a * theta + b) % P for a, b in zip(acc, vals)]
            return acc
        tv = [((a + beta) % P) * ((b + gamma) % P) % P for a, b in zip(compress_coset(ins), compress_coset(tabs))]
        h = lookup_h_terms(dom, dom.coeff_to_extended(z_poly), dom.coeff_to_extended(pi_poly), dom.coeff_to_extended(pt_poly), tv,
                           pk.l0, pk.l_last, pk.l_active_row, beta, gamma, y, h)
    for (b_poly, _b0, f_coeff, _a0) in logd:  # :533-548
        b_coset = dom.coeff_to_extended(b_poly)
        f_coset = dom.coeff_to_extended(f_coeff)
        h = [
            (hv * y + (bc * ((fc * la + beta) % P) - 1)) % P
            for hv, bc, fc, la in zip(h, b_coset, f_coset, pk.l_active_row)
        ]
    out.polys["h_ext"] = h

    # ---- vanishing construct (vanishing/prover.rs:69-120) -------------------
    h = dom.divide_by_vanishing_poly(h)
    h_coeff = dom.extended_to_coeff(h)
    h_pieces = [h_coeff[i : i + n] for i in range(0, len(h_coeff) - n + 1, n)]
    for _ in h_pieces:
        fr_random(rng)  # h_blinds
    h_cms = batch_to_affine([msm(piece, params.g) for piece in h_pieces])
    for A in h_cms:
        tr.write_point(A)
    out.points["h"] = h_cms
    out.polys["h_pieces"] = h_pieces

    x = tr.squeeze_challenge_scalar()  # prover.rs:629
    xn = pow(x, n, P)
    out.challenges["x"] = x

    # ---- evaluations (prover.rs:654-719) ------------------------------------
    aq = cs.advice_queries()
    for col, rot in aq:
        tr.write_scalar(eval_polynomial(advice_polys[col], dom.rotate_omega(x, rot)))
    for col, rot in cs.fixed_queries():  # :674-687
        tr.write_scalar(eval_polynomial(pk.fixed_polys[col], dom.rotate_omega(x, rot)))
    # vanishing.evaluate (vanishing/prover.rs:124-153)
    h_poly = [0] * n
    for piece in reversed(h_pieces):
        h_poly = [(a * xn + b) % P for a, b in zip(h_poly, piece)]
    tr.write_scalar(eval_polynomial(random_poly, x))
    for sp in pk.perm_polys:  # permutation::ProvingKey::evaluate (permutation/prover.rs:227-239)
        tr.write_scalar(eval_polynomial(sp, x))
    x_next = dom.rotate_omega(x, 1)
    x_last = dom.rotate_omega(x, -(bf + 1))
    for si, z in enumerate(perm_sets):  # Constructed::evaluate (:243-290)
        tr.write_scalar(eval_polynomial(z, x))
        tr.write_scalar(eval_polynomial(z, x_next))
        if si + 1 < len(perm_sets):
            tr.write_scalar(eval_polynomial(z, x_last))
    x_inv = dom.rotate_omega(x, -1)
    for (z_poly, pi_poly, pt_poly) in plk:  # lookup::Committed::evaluate (lookup/prover.rs:303-340)
        for poly, pt_ in ((z_poly, x), (z_poly, x_next), (pi_poly, x), (pi_poly, x_inv), (pt_poly, x)):
            tr.write_scalar(eval_polynomial(poly, pt_))
    for (b_poly, b0, f_coeff, a_at_zero) in logd:  # static_lookup/prover.rs:360-370
        tr.write_scalar(eval_polynomial(b0, x))
        tr.write_scalar(eval_polynomial(f_coeff, x))
        tr.write_scalar(a_at_zero)

    # ---- multiopen GWC (prover.rs:721-778, gwc/prover.rs:42-91) -------------
    queries = []
    for col, rot in aq:
        queries.append((dom.rotate_omega(x, rot), advice_polys[col]))
    for z in perm_sets:  # Evaluated::open (permutation/prover.rs:294-344)
        queries.append((x, z))
        queries.append((x_next, z))
    for z in reversed(perm_sets[:-1]):
        queries.append((x_last, z))
    for (z_poly, pi_poly, pt_poly) in plk:  # lookup::Evaluated::open (lookup/prover.rs:343-392)
        queries += [(x, z_poly), (x, pi_poly), (x, pt_poly), (x_inv, pi_poly), (x_next, z_poly)]
    for (b_poly, b0, f_coeff, _a) in logd:
        queries.append((x, b0))
        queries.append((x, f_coeff))
    for col, rot in cs.fixed_queries():
        queries.append((dom.rotate_omega(x, rot), pk.fixed_polys[col]))
    for sp in pk.perm_polys:  # permutation::ProvingKey::open (:215-225)
        queries.append((x, sp))
    queries.append((x, h_poly))
    queries.append((x, random_poly))
    if opener == "shplonk":  # ProverSHPLONK (shplonk/prover.rs:120-286): two commitments whatever the point count
        polys = {id(poly): poly for _, poly in queries}
        sq = [(id(poly), pt, eval_polynomial(poly, pt)) for pt, poly in queries]
        shplonk_prove(tr, n, sq, polys, lambda poly: commit_affine(msm(poly, params.g[: len(poly)])))
        out.proof = bytes(tr.proof)
        return out
    assert opener == "gwc"
    v = tr.squeeze_challenge_scalar()
    out.challenges["v"] = v
    groups = []  # gwc.rs:36-61 first-seen order
    for pt, poly in queries:
        for g in groups:
            if g[0] == pt:
                g[1].append(poly)
                break
        else:
            groups.append((pt, [poly]))
    ws = []
    for z, polys in groups:
        acc = [0] * n
        eacc = 0
        pv = 1
        for poly in polys:
            ev = eval_polynomial(poly, z)
            acc = [(a + b * pv) % P for a, b in zip(acc, poly)]
            eacc = (eacc + ev * pv) % P
            pv = pv * v % P
        acc[0] = (acc[0] - eacc) % P  # `&poly_batch - eval_batch` subtracts from the constant term
        wit = kate_division(acc, z)
        w = commit_affine(msm(wit, params.g[: len(wit)]))
        tr.write_point(w)
        ws.append(w)
    out.points["w"] = ws
    out.proof = bytes(tr.proof)
    return out
