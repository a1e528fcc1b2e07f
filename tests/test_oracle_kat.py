"""CPU: pin the oracle against every golden vector / known-answer test the reference holds for
this path."""
from oracle import bn254 as B
from oracle import sha_tables as ST


def test_from_u512_kats():
    # arithmetic/curves/src/bn256/fr.rs:347-367 and fq.rs:331-351
    w = [0xAAAAAAAAAAAAAAAA] * 8
    assert B.from_u512(w, B.R_MOD) == B._limbs(
        [0x7E7140B5196B9E6F, 0x9ABAC9E4157B6172, 0xF04BC41062FD7322, 0x1185FA9C9FEF6326])
    assert B.from_u512(w, B.Q_MOD) == B._limbs(
        [0x1F8905A172AFFA8A, 0xDE45AD177DCF3306, 0xAAA7987907D73AE2, 0x24D349431D468E30])


def test_fr_constants():
    # bn256/fr.rs:29-118 and the self-checks at :320-345
    r = B.R_MOD
    assert (-pow(r, -1, 1 << 64)) % (1 << 64) == B.FR_INV
    assert B.FR_R == B._limbs([0xAC96341C4FFFFFFB, 0x36FC76959F60CD29, 0x666EA36F7879462E, 0x0E0A77C19A07DF2F])
    assert B.FR_R2 == B._limbs([0x1BB8E645AE216DA7, 0x53FE3AB1E35C59E3, 0x8C49833D53BB8085, 0x0216D0B17F4E44A5])
    assert B.FR_R3 == B._limbs([0x5E94D8E1B4BF0040, 0x2A489CBE1CFBB6B8, 0x893CC664A19FCFED, 0x0CF8594B7FCC657C])
    assert (r - 1) % (1 << B.FR_S) == 0 and ((r - 1) >> B.FR_S) % 2 == 1
    assert pow(B.FR_GENERATOR, (r - 1) >> B.FR_S, r) == B.FR_ROOT_OF_UNITY
    assert pow(B.FR_ROOT_OF_UNITY, 1 << B.FR_S, r) == 1 and pow(B.FR_ROOT_OF_UNITY, 1 << (B.FR_S - 1), r) != 1
    assert B.FR_ROOT_OF_UNITY * B.FR_ROOT_OF_UNITY_INV % r == 1
    assert 2 * B.FR_TWO_INV % r == 1
    assert pow(B.FR_GENERATOR, 1 << B.FR_S, r) == B.FR_DELTA
    assert pow(B.FR_ZETA, 3, r) == 1 and B.FR_ZETA != 1


def test_fq_constants_and_curve():
    # bn256/fq.rs:29-91, bn256/curve.rs:66-83,266-270
    q = B.Q_MOD
    assert (-pow(q, -1, 1 << 64)) % (1 << 64) == B.FQ_INV
    assert B.FQ_R == B._limbs([0xD35D438DC58F0D9D, 0x0A78EB28F5C70B3D, 0x666EA36F7879462C, 0x0E0A77C19A07DF2F])
    assert B.FQ_R2 == B._limbs([0xF32CFC5B538AFA89, 0xB5E71911D44501FB, 0x47AB1EFF0A417FF6, 0x06D89F71CAB8351F])
    neg_one = B._limbs([0x68C3488912EDEFAA, 0x8D087F6872AABF4F, 0x51E1A24709081231, 0x2259D6B14729C0FA])
    assert neg_one == (q - 1) * B.FQ_R % q  # NEGATIVE_ONE is stored in Montgomery form (fq.rs:61-66)
    assert pow(B.FQ_ZETA, 3, q) == 1 and B.FQ_ZETA != 1
    assert B.is_on_curve(B.G1_GEN)
    x, y = B.G1_GEN
    assert B.is_on_curve((B.FQ_ZETA * x % q, y))
    assert B.g1_mul(B.G1_GEN, B.R_MOD) is None  # group order


def test_endo_consistency():
    """bn256/curve.rs:266-270 `test_endo_consistency`: g * (-ENDO_BETA) == g.endo(), with endo() = (ZETA_q * x, -y, z)
    (curve.rs:137-143) and ENDO_BETA the raw limbs of curve.rs:78-83 (stored in tests/golden/reference_kats.json)."""
    import json
    import os

    kat = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_kats.json")))["g1_endomorphism"]
    endo_beta = sum(int(h, 16) << (64 * i) for i, h in enumerate(kat["endo_beta_raw_limbs"]))
    x, y = B.G1_GEN
    assert B.g1_mul(B.G1_GEN, (-endo_beta) % B.R_MOD) == (B.FQ_ZETA * x % B.Q_MOD, (-y) % B.Q_MOD)
    assert pow(endo_beta, 3, B.R_MOD) == 1 and endo_beta != 1  # a cube root of unity in Fr, as the identity requires


def test_toy_sha_round_kat():
    """sha-reference/src/lib.rs:53-80: the one KAT of the toy bit-vector round (2-bit words, `+` is XOR)."""
    import json
    import os

    kat = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_kats.json")))["toy_sha_round_2bit"]
    out = ST.toy_sha_round(tuple(list(w) for w in kat["input_a_to_h"]))
    assert [list(w) for w in out] == kat["expected_a_to_h"]


def test_sha_table_kats():
    # sha/src/tables.rs:168-211
    t = ST.create_rot0_table("test")
    assert len(t) == 256 and (0, 0, 0, 0) in t
    assert (0b1100, 0b01, 0b00, 0b0000_0100) in t and (0b1010, 0b01, 0b10, 0b0000_0110) in t
    t = ST.create_rot1_table("test")
    assert len(t) == 256 and (0, 0, 0, 0) in t
    assert (0b1100, 0b01, 0b00, 0b1110_1001) in t and (0b1010, 0b01, 0b10, 0b0001_1101) in t
    t = ST.create_maj_table("test")
    assert len(t) == 256 and (0, 0, 0, 0) in t
    assert (0b1100, 0b01, 0b00, 0) in t and (0b1010, 0b01, 0b10, 0b10) in t
    t = ST.create_ch_table("test")
    assert len(t) == 256 and (0, 0, 0, 0) in t
    assert (0b1100, 0b01, 0b00, 0) in t and (0b1010, 0b01, 0b10, 0) in t
    t = ST.create_decomposition_table("test", 10)
    assert len(t) == 1024 and (0, 0, 0, 0) in t
    assert (0b10_1010_1010, 0b1010, 0b10, 0b10) in t and (0b11_1011_0110, 0b1011, 0b01, 0b10) in t


def test_g_to_lagrange_matches_closed_form():
    """arithmetic.rs:277-301: the inverse FFT over G1 of [s^i]_1 is the Lagrange-basis SRS; the closed form used for
    setup (kzg/commitment.rs:241-251) must give the same points (the reference's own identity
    commit(lagrange_to_coeff(a)) == commit_lagrange(a), kzg/commitment.rs:570-593, rests on it)."""
    from oracle import kzg

    s = B.fr_random(B.Xoshiro256ss(5))
    for k in (0, 1, 4):
        p = kzg.ParamsKZG(k, s)
        assert kzg.g_to_lagrange(p.g, k) == p.g_lagrange
