"""Synthetic general-PLONK circuits shared by the oracle tests (CPU) and the parity tests (GPU).

The reference has no circuit with golden outputs for gates + copy constraints (its `plonk_api.rs`
test draws from OsRng and only checks acceptance), so these small circuits are built here: an
add/mul chain with next/prev rotations, a scaled-constant gate, an instance column, a fixed column
inside the permutation, optionally a degree-5 gate (-> extended domain 4n, 3 columns per permutation
set) and optionally a static (CQ) lookup next to them.
"""
import numpy as np

from oracle import bn254 as B
from oracle import cq_prover as CP
from oracle import kzg
from oracle import plonk as PL

P = B.R_MOD
A, F, I = PL.ADVICE, PL.FIXED, PL.INSTANCE
TABLE = [0, 1, 6, 8, 10, 12, 14, 16, 18, 20, 22, 24, 26, 28, 30, 32]


def chain_circuit(k, degree5=False, with_lookup=False, lookup_expr=False, plookup=False, phases=False, seed=1):
    """Returns dict(circuit, fixed, advice, instances, mapping, tables)."""
    n = 1 << k
    q_add, q_mul, q_next, q_prev, q_fix, kc = range(6)
    num_fixed = 6
    a, b, c = PL.adv(0), PL.adv(1), PL.adv(2)
    gates = [
        PL.mul(PL.fix(q_add), PL.sub(PL.add(a, b), c)),
        PL.mul(PL.fix(q_mul), PL.sub(PL.mul(a, b), c)),
        PL.mul(PL.fix(q_next), PL.sub(PL.adv(0, 1), c)),
        PL.mul(PL.fix(q_prev), PL.sub(a, PL.adv(2, -1))),
        PL.mul(PL.fix(q_fix), PL.sub(PL.sub(PL.scale(a, 3), PL.fix(kc)), PL.const(5))),
    ]
    if degree5:
        q_pow = num_fixed
        num_fixed += 1
        gates.append(PL.mul(PL.fix(q_pow), PL.sub(PL.mul(PL.mul(a, a), PL.mul(a, a)), c)))
    with_lookup = with_lookup or lookup_expr
    num_advice = 4 if with_lookup else 3
    lookups = [[(3, "t")]] if with_lookup else []
    if lookup_expr:  # input = selector * (advice - advice@prev + 6): degree 2 -> static lookup degree 4 (static_lookup.rs:181-190)
        q_lk = num_fixed
        num_fixed += 1
        lookups = [[(PL.mul(PL.fix(q_lk), PL.add(PL.sub(PL.adv(3), PL.adv(3, -1)), PL.const(6))), "t")]]
    perm_columns = [(A, 0), (A, 1), (A, 2), (I, 0), (F, kc)]
    plookups = []
    if plookup:
        # legacy lookup (plonk/lookup.rs): the pair (q_pl * p, q_pl * (2 p + 1)) must be a row of the fixed table
        # (t0, t1); rows with q_pl = 0 look up (0, 0), which the table holds as its padding row
        num_advice += 1
        pcol = num_advice - 1
        q_pl, t0, t1 = num_fixed, num_fixed + 1, num_fixed + 2
        num_fixed += 3
        plookups = [([PL.mul(PL.fix(q_pl), PL.adv(pcol)), PL.mul(PL.fix(q_pl), PL.add(PL.scale(PL.adv(pcol), 2), PL.const(1)))],
                     [PL.fix(t0), PL.fix(t1)])]
    advice_phases, challenge_phases = None, []
    if phases:
        # three phases (circuit.rs FirstPhase..ThirdPhase): ph1 = c0 * a in phase 1, ph2 = ph1 + c1 in phase 2, with
        # c0 squeezed after the first-phase commitments and c1 after the second-phase ones (prover.rs:436-463)
        ph1, ph2 = num_advice, num_advice + 1
        num_advice += 2
        advice_phases = [0] * (num_advice - 2) + [1, 2]
        challenge_phases = [0, 1]
        gates.append(PL.mul(PL.fix(q_add), PL.sub(PL.adv(ph1), PL.mul(PL.chal(0), a))))
        gates.append(PL.mul(PL.fix(q_add), PL.sub(PL.sub(PL.adv(ph2), PL.adv(ph1)), PL.chal(1))))
    circuit = CP.CqCircuit(k, num_advice, lookups, num_fixed, 1, gates, perm_columns, None, plookups, advice_phases, challenge_phases)
    bf = circuit.blinding_factors()
    u = n - (bf + 1)
    R = u - 2  # chain rows
    rng = B.Xoshiro256ss(seed)
    fixed = [[0] * n for _ in range(num_fixed)]
    adv = [[0] * u for _ in range(num_advice)]
    asm = PL.Assembly(n, perm_columns)
    av, bv = B.fr_random(rng), B.fr_random(rng)
    for r in range(R):
        kind = "pow" if (degree5 and r % 5 == 4) else ("mul" if r % 2 else "add")
        if kind == "add":
            cv = (av + bv) % P
            fixed[q_add][r] = 1
        elif kind == "mul":
            cv = av * bv % P
            fixed[q_mul][r] = 1
        else:
            cv = pow(av, 4, P)
            fixed[q_pow][r] = 1
        adv[0][r], adv[1][r], adv[2][r] = av, bv, cv
        if r + 1 < R:
            fixed[q_next][r] = 1
            fixed[q_prev][r + 1] = 1
            asm.copy((A, 1), r + 1, (A, 0), r)  # b[r+1] == a[r]
            av, bv = cv, av
    # a fixed-constant row: kc = 3a - 5 on row 0
    fixed[q_fix][0] = 1
    fixed[kc][0] = (3 * adv[0][0] - 5) % P
    fixed[kc][1] = adv[1][0]
    asm.copy((F, kc), 1, (A, 1), 0)  # b[0] equals a constant
    instances = [[adv[0][0], adv[2][R - 1]]]
    asm.copy((I, 0), 0, (A, 0), 0)
    asm.copy((I, 0), 1, (A, 2), R - 1)
    asm.copy((A, 2), 3, (A, 2), 3)  # self copy: no-op (keygen.rs:75-77)
    if lookup_expr:
        # d[r] - d[r-1] + 6 in the table on selected rows; unselected rows (and row 0, whose `prev` is a blinding row)
        # look up 0 * anything = 0, which the table holds
        cur = 1000
        for r in range(u):
            if r and r % 3:
                fixed[q_lk][r] = 1
                cur = (cur + TABLE[(r * 5 + 1) % len(TABLE)] - 6) % P
            else:
                cur = (cur * 31 + r) % P
            adv[3][r] = cur
    elif with_lookup:
        for r in range(u):
            adv[3][r] = TABLE[(r * 7 + 3) % len(TABLE)]
    if phases:
        a_col, sel = list(adv[0]), list(fixed[q_add])
        adv[ph1] = lambda ch: [ch[0] * a_col[r] % P if sel[r] else 0 for r in range(u)]
        adv[ph2] = lambda ch: [(ch[0] * a_col[r] + ch[1]) % P if sel[r] else 0 for r in range(u)]
    if plookup:
        vals = [3, 5, 8, 13, 21, 34, 55]
        for i, v_ in enumerate(vals):  # table rows (v, 2v + 1); the remaining rows stay (0, 0)
            fixed[t0][1 + i], fixed[t1][1 + i] = v_, 2 * v_ + 1
        for r in range(u):
            if r % 4 != 3:
                fixed[q_pl][r] = 1
                adv[pcol][r] = vals[(r * 5 + 2) % len(vals)]
            else:
                adv[pcol][r] = 777 + r  # unconstrained
    return dict(circuit=circuit, fixed=fixed, advice=adv, instances=instances, mapping=asm.mapping,
                tables={"t": TABLE} if with_lookup else {})


def plonk_api_circuit(k=5, legacy_lookup=True):
    """`MyCircuit` of halo2_proofs/tests/plonk_api.rs:270-400 (StandardPlonk: "Combined add-mult" and
    "Public input" gates, the legacy `meta.lookup` of column a into the table column sl (:316-319), 12
    equality-enabled columns, 10 multiply/add pairs with two copies each, public input 2,
    a = 2834758237 * ZETA, table [2, a, a, 0] :401-413).  Column allocation order as in `configure` (:281-298);
    rows as SimpleFloorPlanner lays the one-row regions out in call order; the table column is filled with its
    first value from the first unused row on (`assign_table`, circuit/floor_planner/single_pass.rs:183-217)."""
    n = 1 << k
    e, a, b, c, d = range(5)
    sf, sm, sa, sb, sc, sp, sl = range(7)
    A_, F_, I_ = A, F, I
    # queries in registration order: enable_equality(a,b,c); gates; enable_equality(sf,e,d,p,sm,sa,sb,sc,sp)
    gate1 = PL.add(
        PL.sub(PL.add(PL.add(PL.mul(PL.adv(a), PL.fix(sa)), PL.mul(PL.adv(b), PL.fix(sb))),
                      PL.mul(PL.mul(PL.adv(a), PL.adv(b)), PL.fix(sm))),
               PL.mul(PL.adv(c), PL.fix(sc))),
        PL.mul(PL.fix(sf), PL.mul(PL.adv(d, 1), PL.adv(e, -1))))
    gate2 = PL.mul(PL.fix(sp), PL.sub(PL.adv(a), PL.inst(0)))
    perm_columns = [(A_, a), (A_, b), (A_, c), (F_, sf), (A_, e), (A_, d), (I_, 0), (F_, sm), (F_, sa), (F_, sb), (F_, sc), (F_, sp)]
    queries = {
        A_: [(a, 0), (b, 0), (c, 0), (d, 1), (e, -1), (e, 0), (d, 0)],
        # `meta.lookup` queries the table column before the gates are created (circuit.rs:1533-1557)
        F_: ([(sl, 0)] if legacy_lookup else []) + [(sf, 0), (sa, 0), (sb, 0), (sc, 0), (sm, 0), (sp, 0)],
        I_: [(0, 0)],
    }
    plookups = [([PL.adv(a)], [PL.fix(sl)])] if legacy_lookup else []
    nfixed = 7 if legacy_lookup else 6
    circuit = CP.CqCircuit(k, 5, [], nfixed, 1, [gate1, gate2], perm_columns, queries, plookups)
    u = n - (circuit.blinding_factors() + 1)
    fixed = [[0] * n for _ in range(nfixed)]
    adv = [[0] * u for _ in range(5)]
    asm = PL.Assembly(n, perm_columns)
    av = 2834758237 * B.FR_ZETA % P
    adv[a][0] = 2  # public_input: a = 1 + 1, sp = 1
    fixed[sp][0] = 1
    row = 1
    for _ in range(10):
        a2 = av * av % P
        # raw_multiply (:95-153)
        adv[a][row], adv[d][row], adv[b][row], adv[e][row], adv[c][row] = av, pow(av, 4, P), av, pow(av, 4, P), a2
        fixed[sc][row], fixed[sm][row] = 1, 1
        # raw_add (:154-212)
        r2 = row + 1
        adv[a][r2], adv[d][r2], adv[b][r2], adv[e][r2], adv[c][r2] = av, pow(av, 4, P), a2, pow(a2, 4, P), (a2 + av) % P
        fixed[sa][r2], fixed[sb][r2], fixed[sc][r2] = 1, 1, 1
        for _twice in range(2):  # `copy` constrains the pair twice (:213-226)
            asm.copy((A_, a), row, (A_, a), r2)
        for _twice in range(2):
            asm.copy((A_, b), r2, (A_, c), row)
        row += 2
    if legacy_lookup:
        table = [2, av, av, 0]
        for r in range(u):
            fixed[sl][r] = table[r] if r < len(table) else table[0]
    return dict(circuit=circuit, fixed=fixed, advice=adv, instances=[[2]], mapping=asm.mapping, tables={})


class _Params:
    """ParamsKZG look-alike (k, n, g, g_lagrange) whose points were multiplied out by the C oracle."""


def c_scalar_muls(scalars):
    """[scalar * G]_1 as affine tuples via the C restatement (oracle/cq_oracle.c `cqo_g1_mul`): the Python
    double-and-add is too slow beyond a few hundred points."""
    from oracle import cbind as OC

    gen = B.points_to_mont_limbs([B.G1_GEN])[0]
    sm = B.to_mont_limbs(scalars)
    jac = np.stack([OC.g1_mul(gen, sm[i]) for i in range(len(scalars))])
    return B.jac_from_mont_limbs(jac)


def fast_params(k, s):
    """`ParamsKZG::setup_from_toxic_waste` (kzg/commitment.rs:209-276) with the closed-form Lagrange scalars
    (:241-251), points from the C oracle."""
    n = 1 << k
    p = _Params()
    p.k, p.n = k, n
    p.g = c_scalar_muls([pow(s, i, P) for i in range(n)])
    root = kzg._root_for(k)
    mult = (pow(s, n, P) - 1) * B.inv_mod(n % P, P) % P
    sc = []
    rp = 1
    for _ in range(n):
        sc.append(mult * rp % P * B.inv_mod((s - rp) % P, P) % P)
        rp = rp * root % P
    p.g_lagrange = c_scalar_muls(sc)
    return p


def c_msm(coeffs, bases, _cache={}):
    """`best_multiexp` through the C restatement, for oracle proofs at sizes the Python Pippenger cannot reach."""
    from oracle import cbind as OC

    key = (id(bases), len(bases))
    if key not in _cache:
        _cache[key] = (bases, B.points_to_mont_limbs(bases))  # keeps `bases` alive so the id stays unique
    pts = _cache[key][1]
    out = OC.best_multiexp(B.to_mont_limbs(coeffs), pts[: len(coeffs)])
    zs = B.from_mont_limbs(out.reshape(1, 12)[:, 8:12], B.Q_MOD)[0]
    if zs == 0:
        return B.JAC_ID
    xs = B.from_mont_limbs(out.reshape(1, 12)[:, 0:4], B.Q_MOD)[0]
    ys = B.from_mont_limbs(out.reshape(1, 12)[:, 4:8], B.Q_MOD)[0]
    return (xs, ys, zs)


def oracle_env(k, s_seed=0x6371, builder=None, fast=False, **kw):
    """Oracle-side params / pk for `chain_circuit` (SRS layout of tests/my_test.rs:179-205: table SRS over the
    table-sized domain, b0_g1_bound = [s^(n+1+i)]_1, srs_g1_len = 2n).  fast: SRS points from the C oracle."""
    fx = (builder or chain_circuit)(k, **kw)
    s = B.fr_random(B.Xoshiro256ss(s_seed))
    n = 1 << k
    if fast:
        params = fast_params(k, s)
        tsrs = kzg.TableSRS(len(TABLE) - 1, s)
        tabs = {name: kzg.StaticTableValues(v, tsrs.g1) for name, v in fx["tables"].items()}
        b0 = c_scalar_muls([pow(s, n + 1 + i, P) for i in range(n - 1)])
        pk = CP.keygen_pk(fx["circuit"], tabs, tsrs, b0, 424242, fixed=fx["fixed"], perm_mapping=fx["mapping"])
        fx.update(s=s, params=params, pk=pk, tsrs=tsrs, tabs=tabs, srs_g1_len=2 * n)
        return fx
    params = kzg.ParamsKZG(k, s)
    tsrs = kzg.TableSRS(len(TABLE) - 1, s)
    tabs = {name: kzg.StaticTableValues(v, tsrs.g1) for name, v in fx["tables"].items()}
    b0 = kzg._powers_g1(s, 2 * n)[n + 1:]
    pk = CP.keygen_pk(fx["circuit"], tabs, tsrs, b0, 424242, fixed=fx["fixed"], perm_mapping=fx["mapping"])
    fx.update(s=s, params=params, pk=pk, tsrs=tsrs, tabs=tabs, srs_g1_len=2 * n)
    return fx


def to_backend_cs(circuit, gtables):
    """The same constraint system through the backend's `ConstraintSystem` mirror (queries pinned to the
    oracle circuit's registration order)."""
    from sha2_on_cq_halo2_amd import plonk as GP

    cs = GP.ConstraintSystem()
    cols = {A: [cs.advice_column(circuit.phase_of(c_)) for c_ in range(circuit.num_advice)],
            F: [cs.fixed_column() for _ in range(circuit.num_fixed)],
            I: [cs.instance_column() for _ in range(circuit.num_instance)]}

    def conv(e):
        t = e[0]
        if t == "const":
            return GP.Expression.constant(e[1])
        if t == "challenge":
            return GP.Expression.challenge(e[1])
        if t in (A, F, I):
            return cs.query_any(cols[t][e[1]], e[2])
        if t == "neg":
            return -conv(e[1])
        if t == "add":
            return conv(e[1]) + conv(e[2])
        if t == "mul":
            return conv(e[1]) * conv(e[2])
        if t == "scale":
            return conv(e[1]) * int(e[2])
        raise ValueError(t)

    for ph in circuit.challenge_phases:
        cs.challenge_usable_after(ph)
    for kind, idx in circuit.perm_columns:
        cs.enable_equality(cols[kind][idx])
    for li, (ins, tabs) in enumerate(circuit.plookups):
        cs.lookup(f"pl{li}", [(conv(i), conv(t)) for i, t in zip(ins, tabs)])
    for gi, g in enumerate(circuit.gates):
        cs.create_gate(f"g{gi}", [conv(g)])
    for li, lk in enumerate(circuit.lookups):
        cs.lookup_static(f"l{li}", [(cs.query_any(cols[A][c], 0) if isinstance(c, int) else conv(c), gtables[t]) for c, t in lk])
    if circuit.queries is not None:  # registration order pinned by the fixture (e.g. the plonk_api.rs `configure`)
        cs.advice_queries, cs.fixed_queries, cs.instance_queries = (list(circuit.queries[kk]) for kk in (A, F, I))
    assert cs.advice_queries == circuit.advice_queries()
    assert cs.fixed_queries == circuit.fixed_queries()
    assert cs.instance_queries == circuit.instance_queries()
    assert cs.degree() == circuit.degree() and cs.blinding_factors() == circuit.blinding_factors()
    return cs


def random_circuit(k, seed):
    """A random constraint system with a random (unsatisfying) witness: `create_proof` never checks satisfaction, so
    oracle and backend must still agree byte for byte -- a fuzz of the expression interpreter, query bookkeeping,
    rotation handling, permutation chunking and opening-point grouping.  Lookups use inputs that are in their tables
    by construction (legacy: table expression == input expression; static: a column filled from TABLE)."""
    rng = B.Xoshiro256ss(0xF00D + seed)
    rnd = lambda m: rng.next_u64() % m
    n = 1 << k
    num_advice, num_fixed, num_instance = 2 + rnd(3), 1 + rnd(3), rnd(2)
    static_col = None
    if rnd(2):
        static_col = num_advice
        num_advice += 1

    def leaf():
        t = rnd(10)
        if t < 5:
            return PL.adv(rnd(num_advice if static_col is None else num_advice - 1), rnd(5) - 2)
        if t < 8:
            return PL.fix(rnd(num_fixed), rnd(3) - 1)
        if t < 9 and num_instance:
            return PL.inst(0, rnd(2))
        return PL.const(rnd(1 << 20))

    def expr(depth, max_deg):
        if depth == 0 or rnd(4) == 0:
            return leaf()
        t = rnd(6)
        if t == 0:
            return PL.neg(expr(depth - 1, max_deg))
        if t == 1:
            return PL.scale(expr(depth - 1, max_deg), rng.next_u64())
        if t in (2, 3):
            return PL.add(expr(depth - 1, max_deg), expr(depth - 1, max_deg))
        a_ = expr(depth - 1, max_deg)
        da = PL.expr_degree(a_)
        if da >= max_deg:
            return a_
        b_ = expr(depth - 1, max_deg - da)
        return PL.mul(a_, b_) if PL.expr_degree(b_) <= max_deg - da else PL.add(a_, b_)

    gates = [expr(3, 4) for _ in range(1 + rnd(3))]
    cols_all = [(A, i) for i in range(num_advice if static_col is None else num_advice - 1)] + [(F, i) for i in range(num_fixed)] + \
               [(I, i) for i in range(num_instance)]
    perm_columns = [c for c in cols_all if rnd(2)][:5]
    plookups = []
    if rnd(2):
        e1, e2 = expr(2, 2), expr(1, 1)
        plookups = [([e1, e2], [e1, e2])]
    lookups = [[(static_col, "t")]] if static_col is not None else []
    circuit = CP.CqCircuit(k, num_advice, lookups, num_fixed, num_instance, gates, perm_columns, None, plookups)
    # keep at most two distinct rotations per advice column modest: blinding_factors must leave usable rows
    u = n - (circuit.blinding_factors() + 1)
    assert u >= 4
    fixed = [[B.fr_random(rng) for _ in range(n)] for _ in range(num_fixed)]
    adv = [[B.fr_random(rng) for _ in range(u)] for _ in range(num_advice)]
    if static_col is not None:
        adv[static_col] = [TABLE[rnd(len(TABLE))] for _ in range(u)]
    instances = [[B.fr_random(rng) for _ in range(1 + rnd(3))] for _ in range(num_instance)]
    asm = PL.Assembly(n, perm_columns)
    for _ in range(rnd(12) if perm_columns else 0):
        asm.copy(perm_columns[rnd(len(perm_columns))], rnd(u), perm_columns[rnd(len(perm_columns))], rnd(u))
    return dict(circuit=circuit, fixed=fixed, advice=adv, instances=instances, mapping=asm.mapping,
                tables={"t": TABLE} if static_col is not None else {})
