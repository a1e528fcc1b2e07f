"""CPU: the oracle's general-PLONK `create_proof` (gates, fixed/instance columns, permutation argument,
optionally next to a static lookup).  No reference test holds golden bytes for these (plonk_api.rs draws
from OsRng), so the oracle is pinned from the verifier side: every proof it makes must satisfy the
restated verifier equations (plonk/verifier.rs, permutation/verifier.rs, gwc/verifier.rs), and proofs of
violated gates / copy constraints / wrong public inputs must not."""
import pytest

from oracle import bn254 as B
from oracle import cq_prover as CP
from oracle import cq_verifier as CV
from oracle import plonk as PL

from tests.plonk_fixtures import TABLE, oracle_env, plonk_api_circuit

P = B.R_MOD


def _vk(fx):
    params, pk = fx["params"], fx["pk"]
    fixed_cm = B.batch_to_affine([params.commit_lagrange(c) for c in pk.fixed_values])
    perm_cm = B.batch_to_affine([params.commit_lagrange(c) for c in pk.permutations])
    return fixed_cm, perm_cm


def _verify(fx, proof, instances=None, opener="gwc"):
    fixed_cm, perm_cm = _vk(fx)
    return CV.verify_proof(proof, fx["circuit"], 424242, fx["s"], fx["tables"], len(TABLE), fx["srs_g1_len"],
                           instances=fx["instances"] if instances is None else instances,
                           fixed_commitments=fixed_cm, perm_commitments=perm_cm, opener=opener)


def test_assembly_cycles():
    """permutation/keygen.rs:43-112: merged cycles stay single cycles, mapping stays a permutation."""
    cols = [(PL.ADVICE, 0), (PL.ADVICE, 1)]
    asm = PL.Assembly(4, cols)
    asm.copy(cols[0], 0, cols[1], 1)
    asm.copy(cols[1], 1, cols[1], 3)
    asm.copy(cols[0], 0, cols[1], 3)  # already in the same cycle
    flat = [c for col in asm.mapping for c in col]
    assert sorted(flat) == [(i, j) for i in range(2) for j in range(4)]
    cyc, cur = [], (0, 0)
    while cur not in cyc:
        cyc.append(cur)
        cur = asm.mapping[cur[0]][cur[1]]
    assert sorted(cyc) == [(0, 0), (1, 1), (1, 3)]
    with pytest.raises(IndexError):
        asm.copy(cols[0], 4, cols[1], 0)


@pytest.mark.parametrize("kw", [dict(), dict(degree5=True), dict(with_lookup=True), dict(lookup_expr=True), dict(plookup=True),
                                dict(plookup=True, with_lookup=True), dict(phases=True)],
                         ids=["deg3", "deg5", "lookup", "lookup-expr", "plookup", "plookup+cq", "phases"])
def test_plonk_proof_verifies(kw):
    fx = oracle_env(5, **kw)
    cs = fx["circuit"]
    assert cs.degree() == (5 if kw.get("degree5") else 5 if kw.get("plookup") else 4 if kw.get("lookup_expr") else 3)
    assert cs.blinding_factors() == 5  # at most two distinct queries per advice column (circuit.rs:2022-2047)
    tr = CP.create_proof(fx["params"], fx["pk"], fx["advice"], B.Xoshiro256ss(7), instances=fx["instances"])
    chunk = cs.degree() - 2
    sets = -(-len(cs.perm_columns) // chunk)
    L, PLK = len(cs.lookups), len(cs.plookups)
    points = cs.num_advice + 2 * L + 3 * PLK + sets + 5 * L + 1 + (cs.degree() - 1) + 3  # + one W per distinct point
    scalars = (len(cs.advice_queries()) + len(cs.fixed_queries()) + 1 + len(cs.perm_columns) + 3 * sets - 1 + 3 * L + 5 * PLK)
    x_last_used = sets > 1
    assert len(tr.proof) == 32 * (points + scalars + (1 if x_last_used else 0))
    assert _verify(fx, tr.proof)
    # wrong public input
    bad_inst = [[fx["instances"][0][0], (fx["instances"][0][1] + 1) % P]]
    assert not _verify(fx, tr.proof, bad_inst)
    # tampered evaluation
    bad = bytearray(tr.proof)
    bad[-40] ^= 1
    try:
        assert not _verify(fx, bytes(bad))
    except ValueError:
        pass


def test_plonk_unsatisfied_witness_is_rejected():
    fx = oracle_env(5)
    # gate violated
    adv = [list(c) for c in fx["advice"]]
    adv[2][4] = (adv[2][4] + 1) % P
    tr = CP.create_proof(fx["params"], fx["pk"], adv, B.Xoshiro256ss(7), instances=fx["instances"])
    assert not _verify(fx, tr.proof)
    # copy constraint violated, gates intact: b[0] is tied to the fixed constant by the permutation only
    adv = [list(c) for c in fx["advice"]]
    adv[1][0] = (adv[1][0] + 5) % P
    adv[2][0] = (adv[0][0] + adv[1][0]) % P  # keep the add gate of row 0 satisfied
    fixed_rows_next = fx["fixed"][2][0]
    assert fixed_rows_next == 1
    tr = CP.create_proof(fx["params"], fx["pk"], adv, B.Xoshiro256ss(7), instances=fx["instances"])
    assert not _verify(fx, tr.proof)


def test_plonk_api_shape_verifies():
    """The reference's general end-to-end circuit (halo2_proofs/tests/plonk_api.rs `MyCircuit`): two gates, the
    legacy lookup of column a into a 4-value table, 12 permutation columns at degree 4 -> 6 product sets chained
    through z_i(w^last X), queries at next / cur / prev, public input 2.  As in the reference test, acceptance is
    the assertion."""
    fx = oracle_env(5, builder=plonk_api_circuit)
    cs = fx["circuit"]
    assert cs.degree() == 4 and cs.blinding_factors() == 5
    tr = CP.create_proof(fx["params"], fx["pk"], fx["advice"], B.Xoshiro256ss(3), instances=fx["instances"])
    assert _verify(fx, tr.proof)
    assert not _verify(fx, tr.proof, [[3]])  # wrong public input (plonk_api.rs:495-520 expects a failure too)


def test_lagrange_interpolate_and_intermediate_sets():
    """arithmetic.rs:425-478 / shplonk.rs:56-133 on a hand-made query list (the reference's proptest checks the
    same grouping property: commitments with equal point sets share a rotation set)."""
    from oracle import shplonk as SH
    from oracle.poly import eval_polynomial

    rng = B.Xoshiro256ss(5)
    pts = [B.fr_random(rng) for _ in range(4)]
    evs = [B.fr_random(rng) for _ in range(4)]
    poly = SH.lagrange_interpolate(pts, evs)
    assert len(poly) == 4 and all(eval_polynomial(poly, p) == e for p, e in zip(pts, evs))
    assert SH.lagrange_interpolate(pts[:1], evs[:1]) == [evs[0]]
    q = [("a", 1, 10), ("b", 1, 11), ("a", 2, 12), ("c", 2, 13), ("c", 1, 14), ("d", 3, 15)]
    sets, sup = SH.construct_intermediate_sets(q)
    assert sup == [1, 2, 3]
    assert sets == [([1, 2], [("a", [10, 12]), ("c", [14, 13])]), ([1], [("b", [11])]), ([3], [("d", [15])])]


@pytest.mark.parametrize("kw", [dict(), dict(with_lookup=True)], ids=["deg3", "lookup"])
def test_shplonk_proof_verifies(kw):
    """ProverSHPLONK / VerifierSHPLONK (poly/kzg/multiopen/shplonk) on the same circuits: 2 commitments replace
    the per-point witnesses, and the restated verifier equation accepts; tampering is rejected."""
    fx = oracle_env(5, **kw)
    gwc = CP.create_proof(fx["params"], fx["pk"], fx["advice"], B.Xoshiro256ss(7), instances=fx["instances"])
    tr = CP.create_proof(fx["params"], fx["pk"], fx["advice"], B.Xoshiro256ss(7), instances=fx["instances"], opener="shplonk")
    assert len(tr.proof) == len(gwc.proof) - 32 * 2  # 4 GWC witnesses (x, wx, w^-1 x, w^-(bf+1) x) -> 2 points
    assert tr.proof[:-64] == gwc.proof[:-128]  # everything before the multi-open is shared
    assert _verify(fx, tr.proof, opener="shplonk")
    assert not _verify(fx, gwc.proof, opener="shplonk")
    for pos in (len(tr.proof) - 100, len(tr.proof) - 20, 40):
        bad = bytearray(tr.proof)
        bad[pos] ^= 1
        try:
            assert not _verify(fx, bytes(bad), opener="shplonk")
        except ValueError:
            pass


def test_c_backed_oracle_env_matches_python():
    """The C-accelerated helpers used for larger oracle proofs (SRS points by `cqo_g1_mul`, multiexp by
    `cqo_best_multiexp`) give the same SRS and the same proof bytes as the pure-Python path."""
    from tests.plonk_fixtures import c_msm, fast_params

    fx = oracle_env(5, with_lookup=True)
    fp = fast_params(5, fx["s"])
    assert fp.g == fx["params"].g and fp.g_lagrange == fx["params"].g_lagrange
    fast = oracle_env(5, fast=True, with_lookup=True)
    assert fast["pk"].b0_g1_bound == fx["pk"].b0_g1_bound
    a = CP.create_proof(fx["params"], fx["pk"], fx["advice"], B.Xoshiro256ss(9), instances=fx["instances"])
    b = CP.create_proof(fast["params"], fast["pk"], fast["advice"], B.Xoshiro256ss(9), msm=c_msm, instances=fast["instances"])
    assert a.proof == b.proof
