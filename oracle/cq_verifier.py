"""TEST INFRASTRUCTURE ONLY -- acceptance oracle for the proofs of oracle/cq_prover.py.

Restates `plonk/verifier.rs:34-489`, `plonk/vanishing/verifier.rs`,
`plonk/static_lookup/verifier.rs:117-221` and `poly/kzg/multiopen/gwc/verifier.rs:48-128`
working from the PROOF BYTES alone.  The reference finishes with pairing checks
(`tests/my_test.rs:249-258`); in this harness the toxic waste `s` is known (as in the
reference's own test), so every check `e(A,[s^j]_2) = e(B,[1]_2)` is replaced by the
equivalent G1 equation `s^j * A == B`.  That makes this a sound acceptance test for
test SRSs without a pairing implementation (a real pairing verifier is SURVEY 8f-2).
"""
from __future__ import annotations

import hashlib

from .bn254 import (
    G1_B,
    G1_GEN,
    JAC_ID,
    Q_MOD,
    R_MOD,
    from_bytes_wide,
    g1_neg,
    inv_mod,
    jac_add,
    jac_mul,
    jac_to_affine,
    to_jac,
    to_repr,
)
from .bn254 import FR_DELTA
from .plonk import ADVICE, FIXED, expr_eval
from .poly import EvaluationDomain
from .shplonk import shplonk_verifier_terms

P = R_MOD


def g1_from_bytes(b: bytes):
    """derive/curve.rs:603-627."""
    assert len(b) == 32
    ysign = b[31] >> 7
    xb = bytearray(b)
    xb[31] &= 0x7F
    x = int.from_bytes(xb, "little")
    if x >= Q_MOD:
        raise ValueError("non-canonical x")
    if x == 0 and ysign == 0:
        return None
    rhs = (x * x * x + G1_B) % Q_MOD
    y = pow(rhs, (Q_MOD + 1) // 4, Q_MOD)  # q = 3 mod 4
    if y * y % Q_MOD != rhs:
        raise ValueError("not on curve")
    if (y & 1) != ysign:
        y = Q_MOD - y
    return (x, y)


class Blake2bRead:
    """transcript.rs:82-167."""

    def __init__(self, proof: bytes):
        self.state = hashlib.blake2b(digest_size=64, person=b"Halo2-Transcript")
        self.buf = proof
        self.pos = 0

    def common_scalar(self, s):
        self.state.update(b"\x02" + to_repr(s))

    def read_point(self):
        A = g1_from_bytes(self.buf[self.pos : self.pos + 32])
        self.pos += 32
        if A is None:
            raise ValueError("identity in transcript")
        self.state.update(b"\x01" + to_repr(A[0]) + to_repr(A[1]))
        return A

    def read_scalar(self):
        s = int.from_bytes(self.buf[self.pos : self.pos + 32], "little")
        self.pos += 32
        if s >= P:
            raise ValueError("non-canonical scalar")
        self.common_scalar(s)
        return s

    def squeeze(self):
        self.state.update(b"\x00")
        return from_bytes_wide(self.state.copy().digest())


def _smul(A, k):
    return jac_mul(to_jac(A), k % P)


def _eq(J1, J2):
    return jac_to_affine(J1) == jac_to_affine(J2)


def _jac_neg(J):
    return (J[0], (-J[1]) % Q_MOD, J[2])


class _FinalChecks:
    """The verifier's closing equations, each of the form  prod_i e(P_i, [k_i]_2) == 1.

    pairing=False (default): the toxic waste is known in tests, so the equation is checked in G1 as
    sum_i k_i P_i == O -- sound for test SRSs and fast.  pairing=True: the real thing, with the optimal ate pairing
    of oracle/pairing.py and the G2 elements a setup would publish ([1]_2, [s]_2, [T(s)]_2, [Z_V(s)]_2,
    [s^bound]_2; kzg/commitment.rs:253-256, static_lookup.rs:150-160)."""

    def __init__(self, pairing):
        self.pairing = pairing
        self._g2 = {}

    def holds(self, pairs) -> bool:
        if self.pairing == "c":  # the same equations through the C restatement of the pairing (cq_oracle.c): any k in seconds
            from . import cbind as OC

            return OC.pairing_product([jac_to_affine(J) for J, _ in pairs], [k % P for _, k in pairs])
        if not self.pairing:
            acc = JAC_ID
            for J, k in pairs:
                acc = jac_add(acc, jac_mul(J, k % P))
            return jac_to_affine(acc) is None
        from . import pairing as PR

        lst = []
        for J, k in pairs:
            k %= P
            if k not in self._g2:
                self._g2[k] = PR.g2_mul(k)
            lst.append((jac_to_affine(J), self._g2[k]))
        return PR.pairing_product_is_one(lst)


def _gwc_check(tr, proof, queries, s, fc):
    """gwc/verifier.rs:48-128, unbatched: per point z,  e(W, [s]_2) = e(z W + sum_j v^j (C_j - e_j G), [1]_2)."""
    v = tr.squeeze()
    groups = []
    for q in queries:
        for g in groups:
            if g[0] == q[0]:
                g[1].append(q)
                break
        else:
            groups.append((q[0], [q]))
    ws = [tr.read_point() for _ in groups]
    _u = tr.squeeze()
    if tr.pos != len(proof):
        return None
    ok = True
    for (z, qs), w in zip(groups, ws):
        rhs = JAC_ID
        pv = 1
        for q in qs:
            term = jac_add(q[1], _smul(G1_GEN, (-q[2]) % P))
            rhs = jac_add(rhs, jac_mul(term, pv))
            pv = pv * v % P
        ok &= fc.holds([(to_jac(w), s), (_jac_neg(jac_add(_smul(w, z), rhs)), 1)])
    return ok


def verify_proof(proof: bytes, circuit, vk_repr: int, s: int, tables: dict, table_size: int,
                 srs_g1_len: int, instances=(), fixed_commitments=(), perm_commitments=(), opener="gwc",
                 pairing: bool = False) -> bool:
    """Returns True iff every verifier equation holds.

    tables: id -> list of table values (given order); the committed table polynomial
    `t` interpolates the values in SORTED order (static_lookup.rs:139-146).
    srs_g1_len: the `srs_g1_len` handed to `StaticTableValues::commit` (static_lookup.rs:149).
    instances: the public inputs per instance column; fixed_commitments / perm_commitments: the
    verifying key's `fixed_commitments` and `permutation.commitments` (affine points).
    pairing: check the closing equations with real pairings instead of in G1 (see _FinalChecks): True = the Python
    restatement (oracle/pairing.py, smallest proofs only), "c" = the C restatement (oracle/cq_oracle.c), value-for-value
    equal to the Python one and fast enough for any k.
    """
    cs = circuit
    dom = EvaluationDomain(cs.degree(), cs.k)
    n = dom.n
    bf = cs.blinding_factors()
    tr = Blake2bRead(proof)
    tr.common_scalar(vk_repr)
    for vals in instances:  # verifier.rs:93-101
        for v_ in vals:
            tr.common_scalar(v_ % P)
    advice_cm = [None] * cs.num_advice
    challenges = [None] * len(cs.challenge_phases)
    for phase in range(cs.num_phases()):  # verifier.rs:108-131
        for c_ in range(cs.num_advice):
            if cs.phase_of(c_) == phase:
                advice_cm[c_] = tr.read_point()
        for i_, ph in enumerate(cs.challenge_phases):
            if ph == phase:
                challenges[i_] = tr.squeeze()
    theta = tr.squeeze()
    plk_perm = [(tr.read_point(), tr.read_point()) for _ in cs.plookups]  # lookup/verifier.rs:39-52
    lk1 = [(tr.read_point(), tr.read_point()) for _ in cs.lookups]  # f, m
    beta = tr.squeeze()
    gamma = tr.squeeze()
    chunk_len = cs.degree() - 2
    n_sets = (len(cs.perm_columns) + chunk_len - 1) // chunk_len
    perm_z_cm = [tr.read_point() for _ in range(n_sets)]  # permutation/verifier.rs:37-57
    plk_z = [tr.read_point() for _ in cs.plookups]  # lookup/verifier.rs:55-66
    lk2 = [tuple(tr.read_point() for _ in range(5)) for _ in cs.lookups]  # a, qa, a0, b0, p
    random_cm = tr.read_point()
    y = tr.squeeze()
    h_cms = [tr.read_point() for _ in range(dom.quotient_poly_degree)]
    x = tr.squeeze()
    aq = cs.advice_queries()
    advice_evals = [tr.read_scalar() for _ in aq]
    fq = cs.fixed_queries()
    fixed_evals = [tr.read_scalar() for _ in fq]
    random_eval = tr.read_scalar()
    perm_common_evals = [tr.read_scalar() for _ in cs.perm_columns]  # permutation/verifier.rs:60-70
    perm_evals = []
    for si in range(n_sets):  # :73-105
        ze, zn = tr.read_scalar(), tr.read_scalar()
        zl = tr.read_scalar() if si + 1 < n_sets else None
        perm_evals.append((ze, zn, zl))
    plk_evals = [tuple(tr.read_scalar() for _ in range(5)) for _ in cs.plookups]  # z, z_next, a', a'_inv, s' (:69-92)
    lk_evals = [(tr.read_scalar(), tr.read_scalar(), tr.read_scalar()) for _ in cs.lookups]

    xn = pow(x, n, P)
    # l_i_range(x, xn, -(bf+1)..=0)  (domain.rs:453-478)
    rots = list(range(-(bf + 1), 1))
    common = (xn - 1) * dom.barycentric_weight % P
    l_evals = []
    for r in rots:
        wr = dom.rotate_omega(1, r)
        l_evals.append(dom.rotate_omega(inv_mod((x - wr) % P, P) * common % P, r))
    l_last = l_evals[0]
    l_blind = sum(l_evals[1 : 1 + bf]) % P
    l_0 = l_evals[1 + bf]
    active = (1 - (l_last + l_blind)) % P

    # instance evaluations from the public inputs (verifier.rs:208-248): p(w^r x) = sum_i v_i l_i(w^r x)
    iq = cs.instance_queries()
    instance_evals = []
    for col, rot in iq:
        pt = dom.rotate_omega(x, rot)
        ptn = pow(pt, n, P)
        acc = 0
        for i, v_ in enumerate(instances[col]):
            wi = dom.rotate_omega(1, i)
            li = (ptn - 1) * dom.barycentric_weight % P * wi % P * inv_mod((pt - wi) % P, P) % P
            acc = (acc + v_ * li) % P
        instance_evals.append(acc)

    def query_eval(kind, col, rot):
        if kind == ADVICE:
            return advice_evals[aq.index((col, rot))]
        if kind == FIXED:
            return fixed_evals[fq.index((col, rot))]
        return instance_evals[iq.index((col, rot))]

    # expressions (static_lookup/verifier.rs:182-221), folded by y (vanishing/verifier.rs:105-106)
    h_eval = 0
    for g in cs.gates:  # verifier.rs:300-323
        h_eval = (h_eval * y + expr_eval(g, query_eval, challenges)) % P
    if n_sets:  # permutation/verifier.rs:108-206
        exprs = [l_0 * (1 - perm_evals[0][0]) % P,
                 (perm_evals[-1][0] * perm_evals[-1][0] - perm_evals[-1][0]) * l_last % P]
        for si in range(1, n_sets):
            exprs.append((perm_evals[si][0] - perm_evals[si - 1][2]) * l_0 % P)
        for si in range(n_sets):
            cols = cs.perm_columns[si * chunk_len : (si + 1) * chunk_len]
            sig = perm_common_evals[si * chunk_len : (si + 1) * chunk_len]
            left = perm_evals[si][1]
            for (kind, idx), se in zip(cols, sig):
                left = left * ((query_eval(kind, idx, 0) + beta * se + gamma) % P) % P
            right = perm_evals[si][0]
            cur_delta = beta * x % P * pow(FR_DELTA, si * chunk_len, P) % P
            for kind, idx in cols:
                right = right * ((query_eval(kind, idx, 0) + cur_delta + gamma) % P) % P
                cur_delta = cur_delta * FR_DELTA % P
            exprs.append((left - right) * active % P)
        for e in exprs:
            h_eval = (h_eval * y + e) % P
    for (ins, tabs), (ze, zn, ae, ai, se) in zip(cs.plookups, plk_evals):  # lookup/verifier.rs:96-157
        def compress(exprs):
            acc = 0
            for e in exprs:
                acc = (acc * theta + expr_eval(e, query_eval, challenges)) % P
            return acc
        left = zn * ((ae + beta) % P) % P * ((se + gamma) % P) % P
        right = ze * ((compress(ins) + beta) % P) % P * ((compress(tabs) + gamma) % P) % P
        for e in (l_0 * (1 - ze) % P, l_last * ((ze * ze - ze) % P) % P, (left - right) * active % P, l_0 * (ae - se) % P,
                  (ae - se) * (ae - ai) % P * active % P):
            h_eval = (h_eval * y + e) % P
    beta_inv = inv_mod(beta, P)
    n_inv = inv_mod(n % P, P)
    for (b0_eval, f_eval, a_at_zero) in lk_evals:
        b_at_zero = (table_size * a_at_zero + (bf + 1) * beta_inv) % P * n_inv % P
        b_eval = (b0_eval * x + b_at_zero) % P
        expr = (b_eval * ((active * f_eval + beta) % P) - 1) % P
        h_eval = (h_eval * y + expr) % P
    expected_h_eval = h_eval * inv_mod((xn - 1) % P, P) % P
    h_commitment = JAC_ID
    for c in reversed(h_cms):
        h_commitment = jac_add(jac_mul(h_commitment, xn), to_jac(c))

    # queries in prover.rs:721-773 / verifier.rs order: (point, commitment, eval, commitment identity)
    queries = []
    for (col, rot), ev in zip(aq, advice_evals):
        queries.append((dom.rotate_omega(x, rot), to_jac(advice_cm[col]), ev, ("advice", col)))
    x_next = dom.rotate_omega(x, 1)
    x_last = dom.rotate_omega(x, -(bf + 1))
    zsets = list(enumerate(zip(perm_z_cm, perm_evals)))
    for si, (zc, (ze, zn, _zl)) in zsets:  # permutation/verifier.rs:208-252
        queries.append((x, to_jac(zc), ze, ("z", si)))
        queries.append((x_next, to_jac(zc), zn, ("z", si)))
    for si, (zc, (_ze, _zn, zl)) in reversed(zsets[:-1]):
        queries.append((x_last, to_jac(zc), zl, ("z", si)))
    x_inv = dom.rotate_omega(x, -1)
    for li, ((a_cm, s_cm), z_cm, (ze, zn, ae, ai, se)) in enumerate(zip(plk_perm, plk_z, plk_evals)):  # lookup/verifier.rs:159-216
        queries += [(x, to_jac(z_cm), ze, ("lz", li)), (x, to_jac(a_cm), ae, ("la", li)), (x, to_jac(s_cm), se, ("ls", li)),
                    (x_inv, to_jac(a_cm), ai, ("la", li)), (x_next, to_jac(z_cm), zn, ("lz", li))]
    for li, ((f_cm, _m), (a, qa, a0, b0, p_), (b0_eval, f_eval, _az)) in enumerate(zip(lk1, lk2, lk_evals)):
        queries.append((x, to_jac(b0), b0_eval, ("b0", li)))
        queries.append((x, to_jac(f_cm), f_eval, ("f", li)))
    for (col, rot), ev in zip(fq, fixed_evals):  # verifier.rs:447-459
        queries.append((dom.rotate_omega(x, rot), to_jac(fixed_commitments[col]), ev, ("fixed", col)))
    for ci, (cm_, ev) in enumerate(zip(perm_commitments, perm_common_evals)):  # permutation/verifier.rs:255-268
        queries.append((x, to_jac(cm_), ev, ("sigma", ci)))
    queries.append((x, h_commitment, expected_h_eval, ("h",)))
    queries.append((x, to_jac(random_cm), random_eval, ("random",)))

    ok = True
    fc = _FinalChecks(pairing)
    if opener == "shplonk":  # shplonk/verifier.rs:54-148
        y_ = tr.squeeze()
        v_ = tr.squeeze()
        h1 = tr.read_point()
        u_ = tr.squeeze()
        h2 = tr.read_point()
        _batch = tr.squeeze()
        if tr.pos != len(proof):
            return False
        terms, r_outer, z_0 = shplonk_verifier_terms([(q[3], q[0], q[2]) for q in queries], y_, v_, u_)
        by_key = {q[3]: q[1] for q in queries}
        rhs = JAC_ID
        for sc_, key in terms:
            rhs = jac_add(rhs, jac_mul(by_key[key], sc_))
        rhs = jac_add(rhs, _smul(G1_GEN, (-r_outer) % P))
        rhs = jac_add(rhs, _smul(h1, (-z_0) % P))
        ok &= fc.holds([(to_jac(h2), s), (_jac_neg(jac_add(rhs, _smul(h2, u_))), 1)])  # e(h2,[s]) = e(rhs + u*h2,[1])
    else:
        ok &= _gwc_check(tr, proof, queries, s, fc)
        if ok is None:
            return False
    # CQ pairings (static_lookup/verifier.rs:138-177), each equation separately
    for lk, (f_cm, m_cm), (a, qa, a0, b0, p_), (_b0e, _fe, a_at_zero) in zip(cs.lookups, lk1, lk2, lk_evals):
        # [T(s)] compressed with theta over sorted-value interpolants (:151-158, static_lookup.rs:139-146)
        ts = 0
        for _, tid in lk:
            vals = sorted(tables[tid])
            tdom = EvaluationDomain(2, len(vals).bit_length() - 1)
            coeffs = tdom.lagrange_to_coeff(vals)
            tv = 0
            for c in reversed(coeffs):
                tv = (tv * s + c) % P
            ts = (ts * theta + tv) % P
        zv = (pow(s, table_size, P) - 1) % P
        # e(a,[T]) = e(qa,[Zv]) e(m - beta a,[1])
        ok &= fc.holds([(to_jac(a), ts), (_jac_neg(to_jac(qa)), zv), (_jac_neg(jac_add(to_jac(m_cm), _smul(a, (-beta) % P))), 1)])
        # e(b0,[s^(L-1-(n-2))]) = e(p,[1])
        ok &= fc.holds([(to_jac(b0), pow(s, srs_g1_len - 1 - (n - 2), P)), (_jac_neg(to_jac(p_)), 1)])
        # e(a - [a(0)],[1]) = e(a0,[s])
        ok &= fc.holds([(jac_add(to_jac(a), _smul(G1_GEN, (-a_at_zero) % P)), 1), (_jac_neg(to_jac(a0)), s)])
    return bool(ok)
