"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the SHPLONK multi-open.

Restates `poly/kzg/multiopen/shplonk.rs:56-133` (`construct_intermediate_sets`),
`shplonk/prover.rs:28-286` and `shplonk/verifier.rs:54-148`, with `lagrange_interpolate` /
`evaluate_vanishing_polynomial` (`arithmetic.rs:425-505`).  The reference keeps point sets in hash sets; every
value derived from them (interpolants, vanishing products, exact divisions) is independent of the iteration
order, so insertion order is used here.

Parity status: "parity unpinned" against Rust (no golden proof; `plonk_api.rs` only checks acceptance); pinned
from the verifier side in `tests/test_oracle_plonk.py`.
"""
from __future__ import annotations

from .bn254 import R_MOD, inv_mod
from .poly import eval_polynomial, kate_division

P = R_MOD


def lagrange_interpolate(points, evals):
    """arithmetic.rs:425-478 (coefficients of the unique polynomial of degree < len(points))."""
    assert len(points) == len(evals)
    if len(points) == 1:
        return [evals[0] % P]
    final = [0] * len(points)
    for j, xj in enumerate(points):
        tmp = [1]
        for k, xk in enumerate(points):
            if k == j:
                continue
            denom = inv_mod((xj - xk) % P, P)
            nxt = [0] * (len(tmp) + 1)
            for i, a in enumerate(tmp + [0]):
                b = ([0] + tmp)[i]
                nxt[i] = (a * ((-denom * xk) % P) + b * denom) % P
            tmp = nxt
        for i, c in enumerate(tmp):
            final[i] = (final[i] + c * evals[j]) % P
    return final


def evaluate_vanishing_polynomial(roots, z):
    """arithmetic.rs:480-505."""
    acc = 1
    for r in roots:
        acc = acc * ((z - r) % P) % P
    return acc


def construct_intermediate_sets(queries):
    """shplonk.rs:56-133.  queries: (commitment key, point, eval).  Returns (rotation_sets, super_point_set)
    with rotation_sets = [(points, [(key, evals aligned with points)])] in first-seen order."""
    super_points = []
    commitment_sets = []  # (key, [points])
    for key, pt, _ in queries:
        if pt not in super_points:
            super_points.append(pt)
        for ck, pts in commitment_sets:
            if ck == key:
                if pt not in pts:
                    pts.append(pt)
                break
        else:
            commitment_sets.append((key, [pt]))

    def get_eval(key, pt):
        for k2, p2, ev in queries:
            if k2 == key and p2 == pt:
                return ev
        raise KeyError

    rotation_sets = []  # (points, [keys])
    for key, pts in commitment_sets:
        for rpts, keys in rotation_sets:
            if set(rpts) == set(pts):
                keys.append(key)
                break
        else:
            rotation_sets.append((pts, [key]))
    return [(pts, [(key, [get_eval(key, pt) for pt in pts]) for key in keys]) for pts, keys in rotation_sets], super_points


def shplonk_prove(tr, n, queries, polys, commit):
    """shplonk/prover.rs:120-286.  queries: (key, point, eval); polys: key -> coefficient list (length n);
    commit(poly) -> affine point.  Writes two points to the transcript."""
    y = tr.squeeze_challenge_scalar()
    rotation_sets, super_points = construct_intermediate_sets(queries)
    v = tr.squeeze_challenge_scalar()
    ext = []  # per rotation set: (points, [(key, low-degree equivalent)])
    for pts, commitments in rotation_sets:
        ext.append((pts, [(key, lagrange_interpolate(pts, evals)) for key, evals in commitments]))
    h_x = [0] * n
    pv = 1
    for pts, commitments in ext:  # quotient_contribution (:141-168)
        n_x = [0] * n
        py = 1
        for key, r in commitments:
            num = list(polys[key]) + [0] * (n - len(polys[key]))
            for i, c in enumerate(r):
                num[i] = (num[i] - c) % P
            n_x = [(a + b * py) % P for a, b in zip(n_x, num)]
            py = py * y % P
        q = n_x
        for pt in pts:  # div_by_vanishing (:28-34)
            q = kate_division(q, pt)
        q = q + [0] * (n - len(q))
        h_x = [(a + b * pv) % P for a, b in zip(h_x, q)]
        pv = pv * v % P
    tr.write_point(commit(h_x))
    u = tr.squeeze_challenge_scalar()
    l_x = [0] * n
    z_diffs = []
    pv = 1
    for pts, commitments in ext:  # linearisation_contribution (:205-224)
        z_i = evaluate_vanishing_polynomial([p_ for p_ in super_points if p_ not in pts], u)
        z_diffs.append(z_i)
        inner = [0] * n
        py = 1
        for key, r in commitments:
            lin = list(polys[key]) + [0] * (n - len(polys[key]))
            lin[0] = (lin[0] - eval_polynomial(r, u)) % P
            inner = [(a + b * py) % P for a, b in zip(inner, lin)]
            py = py * y % P
        l_x = [(a + b * z_i % P * pv) % P for a, b in zip(l_x, inner)]
        pv = pv * v % P
    zt_eval = evaluate_vanishing_polynomial(super_points, u)
    l_x = [(a - b * zt_eval) % P for a, b in zip(l_x, h_x)]
    assert eval_polynomial(l_x, u) == 0  # :251-255
    h2 = kate_division(l_x, u)
    z0_inv = inv_mod(z_diffs[0], P)
    h2 = [c * z0_inv % P for c in h2]
    tr.write_point(commit(h2))
    return y, v, u


def shplonk_verifier_terms(queries, y, v, u):
    """shplonk/verifier.rs:88-138: returns ([(scalar, commitment key)], r_outer_acc, z_0) such that the opening
    holds iff  e(h2,[s]_2) = e(sum scalar*C - r_outer_acc*G - z_0*h1 + u*h2, [1]_2)."""
    rotation_sets, super_points = construct_intermediate_sets(queries)
    terms = []
    r_outer = 0
    z_0 = z_0_diff_inv = 0
    pv = 1
    for i, (pts, commitments) in enumerate(rotation_sets):
        z_diff_i = evaluate_vanishing_polynomial([p_ for p_ in super_points if p_ not in pts], u)
        if i == 0:
            z_0 = evaluate_vanishing_polynomial(pts, u)
            z_0_diff_inv = inv_mod(z_diff_i, P)
            z_diff_i = 1
        else:
            z_diff_i = z_diff_i * z_0_diff_inv % P
        r_inner = 0
        py = 1
        for key, evals in commitments:
            r_eval = py * eval_polynomial(lagrange_interpolate(pts, evals), u) % P
            terms.append((py * pv % P * z_diff_i % P, key))
            r_inner = (r_inner + r_eval) % P
            py = py * y % P
        r_outer = (r_outer + pv * r_inner % P * z_diff_i) % P
        pv = pv * v % P
    return terms, r_outer, z_0
