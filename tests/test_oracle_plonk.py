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

from tests.plonk_fixtures import TABLE, oracle_env

P = B.R_MOD


def _vk(fx):
    params, pk = fx["params"], fx["pk"]
    fixed_cm = B.batch_to_affine([params.commit_lagrange(c) for c in pk.fixed_values])
    perm_cm = B.batch_to_affine([params.commit_lagrange(c) for c in pk.permutations])
    return fixed_cm, perm_cm


def _verify(fx, proof, instances=None):
    fixed_cm, perm_cm = _vk(fx)
    return CV.verify_proof(proof, fx["circuit"], 424242, fx["s"], fx["tables"], len(TABLE), fx["srs_g1_len"],
                           instances=fx["instances"] if instances is None else instances,
                           fixed_commitments=fixed_cm, perm_commitments=perm_cm)


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


@pytest.mark.parametrize("kw", [dict(), dict(degree5=True), dict(with_lookup=True), dict(lookup_expr=True)],
                         ids=["deg3", "deg5", "lookup", "lookup-expr"])
def test_plonk_proof_verifies(kw):
    fx = oracle_env(5, **kw)
    cs = fx["circuit"]
    assert cs.degree() == (5 if kw.get("degree5") else 4 if kw.get("lookup_expr") else 3)
    assert cs.blinding_factors() == 5  # at most two distinct queries per advice column (circuit.rs:2022-2047)
    tr = CP.create_proof(fx["params"], fx["pk"], fx["advice"], B.Xoshiro256ss(7), instances=fx["instances"])
    chunk = cs.degree() - 2
    sets = -(-len(cs.perm_columns) // chunk)
    L = len(cs.lookups)
    points = cs.num_advice + 2 * L + sets + 5 * L + 1 + (cs.degree() - 1) + 3  # + one W per distinct point
    scalars = (len(cs.advice_queries()) + len(cs.fixed_queries()) + 1 + len(cs.perm_columns) + 3 * sets - 1 + 3 * L)
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
