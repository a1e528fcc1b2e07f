"""CPU: the oracle's `create_proof` restatements.  (1) The Python prover's proofs pass the
acceptance verifier (every verifier equation of plonk/verifier.rs / static_lookup/verifier.rs /
gwc/verifier.rs, pairings replaced by s-multiplications) and tampering is rejected.  (2) The C
restatement (the CPU baseline bench.py times) produces the same bytes as the Python one."""
import numpy as np

from oracle import bn254 as B
from oracle import cbind as OC
from oracle import cq_prover as CP
from oracle import cq_verifier as CV
from oracle import kzg

TV = {"table": [0, 1, 6, 8, 10, 12, 14, 16, 18, 20, 22, 24, 26, 28, 30, 32],
      "table_2": [0, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16]}


def _my_test_env():
    """halo2_proofs/tests/my_test.rs:179-259 (K=3, N=16)."""
    k = 3
    s = B.fr_random(B.Xoshiro256ss(0x6371))
    tsrs = kzg.TableSRS(15, s)
    tabs = {n: kzg.StaticTableValues(v, tsrs.g1) for n, v in TV.items()}  # the reference's O(N^2) qs
    params = kzg.ParamsKZG(k, s)
    circ = CP.CqCircuit(k, 2, [[(0, "table"), (1, "table_2")]])
    pk = CP.keygen_pk(circ, tabs, tsrs, tsrs.g1[(1 << k) + 1:], vk_repr=424242)
    return s, params, circ, pk, tsrs, tabs


def test_python_prover_my_test_shape_verifies():
    s, params, circ, pk, tsrs, tabs = _my_test_env()
    # closed-form cached quotients agree with StaticTableValues::new's construction
    assert tabs["table"].qs == kzg.StaticTableValues.qs_closed_form(TV["table"], s)
    tr = CP.create_proof(params, pk, [[30, 6], [15, 3]], B.Xoshiro256ss(7))
    assert len(tr.proof) == 608  # 13 points + 6 scalars
    assert CV.verify_proof(tr.proof, circ, 424242, s, TV, 16, 16)
    for pos in (5, 200, 416 + 3, 600):
        bad = bytearray(tr.proof)
        bad[pos] ^= 1
        try:
            assert not CV.verify_proof(bytes(bad), circ, 424242, s, TV, 16, 16)
        except ValueError:
            pass  # tampered point no longer decodes
    # a lookup of a value outside the table is an error (static_lookup/prover.rs:141)
    try:
        CP.create_proof(params, pk, [[31, 6], [15, 3]], B.Xoshiro256ss(7))
        assert False
    except KeyError:
        pass
    # same row index required across the vector lookup (:148)
    try:
        CP.create_proof(params, pk, [[30, 6], [3, 15]], B.Xoshiro256ss(7))
        assert False
    except ValueError:
        pass


def test_c_prover_matches_python_prover():
    s, params, circ, pk, tsrs, tabs = _my_test_env()
    adv_u = [[30, 6], [15, 3]]
    tr = CP.create_proof(params, pk, adv_u, B.Xoshiro256ss(99))
    n = 8
    advice = [B.to_mont_limbs(c + [0] * (n - len(c))) for c in adv_u]
    la = OC.keygen_l_active(3, circ.blinding_factors())
    assert np.array_equal(la, B.to_mont_limbs(pk.l_active_row))
    proof = OC.create_proof(
        3, 2, [[(0, 0), (1, 1)]],
        [B.to_mont_limbs(TV["table"]), B.to_mont_limbs(TV["table_2"])],
        [B.points_to_mont_limbs(tabs["table"].qs), B.points_to_mont_limbs(tabs["table_2"].qs)],
        B.points_to_mont_limbs(params.g), B.points_to_mont_limbs(params.g_lagrange),
        B.points_to_mont_limbs(tsrs.g1_lagrange), B.points_to_mont_limbs(tsrs.g_lagrange_opening_at_0),
        B.points_to_mont_limbs(pk.b0_g1_bound), la, B.to_mont_limbs([424242])[0], advice, 99)
    assert proof == tr.proof
