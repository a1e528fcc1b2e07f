"""CPU: the Python restatement of the BN254 pairing (oracle/pairing.py) against the properties the reference's own
tests assert (arithmetic/curves/src/bn256/engine.rs:662-762), and the acceptance verifier run with REAL pairing
checks on the reference's CQ end-to-end shape (tests/my_test.rs:249-258) instead of the s-multiplication shortcut."""
from oracle import bn254 as B
from oracle import cq_prover as CP
from oracle import cq_verifier as CV
from oracle import pairing as PR
from tests.test_oracle_prover import TV, _my_test_env


def _g1(k):
    return B.jac_to_affine(B.jac_mul(B.to_jac(B.G1_GEN), k % B.R_MOD))


def test_g2_constants_are_the_references():
    assert PR.is_on_twist(PR.G2_GEN)  # curve.rs:100-129 on y^2 = x^3 + 3/(9+i) (curve.rs:85-98)
    assert PR.B2 * PR.FQ2([9, 1]) == PR.FQ2([3, 0])
    assert PR.ec_mul(PR.G2_GEN, B.R_MOD) is None  # order r
    assert PR.ATE_LOOP_COUNT == 29793968203157093288  # 6 * BN_X + 2 (engine.rs:18-26)


def test_pairing_properties():
    """engine.rs:662-687 `test_pairing` and :717-762 `random_bilinearity_tests` (a few random draws)."""
    g2 = PR.G2_GEN
    e = PR.pairing(B.G1_GEN, g2)
    assert e != PR.FQ12.one() and e ** B.R_MOD == PR.FQ12.one()
    assert PR.pairing(B.G1_GEN, PR.ec_double(g2)) == PR.pairing(_g1(2), g2) == e * e
    assert PR.pairing(B.G1_GEN, PR.ec_mul(g2, 4)) == PR.pairing(_g1(2), PR.ec_double(g2))
    rng = B.Xoshiro256ss(0x5962BE5D)
    for _ in range(2):
        ka, kb, c, d = (B.fr_random(rng) for _ in range(4))
        a, b = _g1(ka), PR.g2_mul(kb)
        ac, ad = _g1(ka * c), _g1(ka * d)
        bc, bd = PR.g2_mul(kb * c), PR.g2_mul(kb * d)
        acbd, adbc = PR.pairing(ac, bd), PR.pairing(ad, bc)
        assert acbd == adbc == PR.pairing(a, b) ** (c * d % B.R_MOD)
    # multi-Miller loop with one final exponentiation (engine.rs:811-840)
    assert PR.pairing_product_is_one([(_g1(5), PR.g2_mul(7)), (_g1(-35), g2)])
    assert not PR.pairing_product_is_one([(_g1(5), PR.g2_mul(7)), (_g1(-34), g2)])
    assert PR.miller_loop(None, B.G1_GEN) == PR.FQ12.one() and PR.miller_loop(g2, None) == PR.FQ12.one()


def test_my_test_shape_accepted_with_real_pairings():
    """tests/my_test.rs:179-259: the proof of the K=3 CQ circuit passes every closing pairing check (GWC opening and
    the three CQ equations), and a tampered proof does not."""
    s, params, circ, pk, tsrs, tabs = _my_test_env()
    tr = CP.create_proof(params, pk, [[30, 6], [15, 3]], B.Xoshiro256ss(7))
    assert CV.verify_proof(tr.proof, circ, 424242, s, TV, 16, 16, pairing=True)
    bad = bytearray(tr.proof)
    bad[200] ^= 1
    try:
        assert not CV.verify_proof(bytes(bad), circ, 424242, s, TV, 16, 16, pairing=True)
    except ValueError:
        pass
    sh = CP.create_proof(params, pk, [[30, 6], [15, 3]], B.Xoshiro256ss(7), opener="shplonk")
    assert CV.verify_proof(sh.proof, circ, 424242, s, TV, 16, 16, opener="shplonk", pairing=True)


def test_c_pairing_equals_the_python_restatement():
    """oracle/cq_oracle.c's pairing section against oracle/pairing.py: the same Gt VALUES (coefficient for coefficient
    in the Fq[w]/(w^12 - 18 w^6 + 82) basis) for random points, bilinearity and the product-is-one check through C, the
    identity cases, and [k]_2 itself."""
    import numpy as np

    from oracle import cbind as OC

    rng = B.Xoshiro256ss(0xC0DE)
    for _ in range(2):
        ka, kb = B.fr_random(rng), B.fr_random(rng)
        ok, val = OC.pairing_product([_g1(ka)], [kb], want_value=True)
        assert not ok and val == PR.pairing(_g1(ka), PR.g2_mul(kb)).c
    # e(aP, bQ) e(-abP, Q) = 1; one unit off is not
    a, b = B.fr_random(rng), B.fr_random(rng)
    assert OC.pairing_product([_g1(a), _g1(-a * b)], [b, 1])
    assert not OC.pairing_product([_g1(a), _g1(-a * b + 1)], [b, 1])
    assert OC.pairing_product([None, _g1(3)], [5, 0])  # identities on either side contribute 1
    out = np.zeros(16, dtype=np.uint64)
    k = np.array([(b >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(4)], dtype=np.uint64)
    assert OC.lib().cqo_g2_mul(k.ctypes.data, out.ctypes.data) == 1
    x, y = PR.g2_mul(b)
    rinv = pow(1 << 256, -1, B.Q_MOD)
    got = [sum(int(out[4 * c + j]) << (64 * j) for j in range(4)) * rinv % B.Q_MOD for c in range(4)]
    assert got == [x.c[0], x.c[1], y.c[0], y.c[1]]


def test_my_test_shape_accepted_with_c_pairings():
    s, params, circ, pk, tsrs, tabs = _my_test_env()
    tr = CP.create_proof(params, pk, [[30, 6], [15, 3]], B.Xoshiro256ss(7))
    assert CV.verify_proof(tr.proof, circ, 424242, s, TV, 16, 16, pairing="c")
    bad = bytearray(tr.proof)
    bad[200] ^= 1
    try:
        assert not CV.verify_proof(bytes(bad), circ, 424242, s, TV, 16, 16, pairing="c")
    except ValueError:
        pass
