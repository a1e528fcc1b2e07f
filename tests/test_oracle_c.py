"""CPU: the plain-C oracle must agree bit-for-bit with the Python big-int oracle (which is pinned
by the reference's KATs), so either can serve as the checker for the HIP path."""
import numpy as np

from oracle import bn254 as B
from oracle import cbind as OC
from oracle import poly as OP
from tests.util import random_points, random_scalars


def _omega(log_n):
    w = B.FR_ROOT_OF_UNITY
    for _ in range(log_n, B.FR_S):
        w = w * w % B.R_MOD
    return w


def test_c_field_mul_matches_python():
    rng = B.Xoshiro256ss(1)
    for _ in range(200):
        a, b = B.fr_random(rng), B.fr_random(rng)
        out = np.zeros(4, dtype=np.uint64)
        am, bm = B.to_mont_limbs([a]), B.to_mont_limbs([b])  # keep the arrays alive across the call
        OC.lib().cqo_fr_mul(am.ctypes.data, bm.ctypes.data, out.ctypes.data)
        assert B.from_mont_limbs(out)[0] == a * b % B.R_MOD
        a %= B.Q_MOD
        b %= B.Q_MOD
        am, bm = B.to_mont_limbs([a], B.Q_MOD), B.to_mont_limbs([b], B.Q_MOD)
        OC.lib().cqo_fq_mul(am.ctypes.data, bm.ctypes.data, out.ctypes.data)
        assert B.from_mont_limbs(out, B.Q_MOD)[0] == a * b % B.Q_MOD


def test_c_fft_matches_python():
    for log_n in (0, 1, 2, 3, 5, 8, 10):
        vals = random_scalars(1 << log_n, 50 + log_n)
        w = _omega(log_n)
        exp = OP.best_fft(list(vals), w, log_n)
        got = OC.best_fft(B.to_mont_limbs(vals), B.to_mont_limbs([w])[0], log_n)
        assert np.array_equal(got, B.to_mont_limbs(exp))


def test_c_multiexp_matches_python():
    for n in (1, 3, 5, 40, 300, 1500):
        sc = random_scalars(n, n)
        pts = random_points(n, n + 1)
        if n > 10:
            sc[0] = 0
            pts[3] = None
        exp = B.jac_to_affine(OP.best_multiexp(sc, pts))
        got = OC.best_multiexp(B.to_mont_limbs(sc), B.points_to_mont_limbs(pts))
        assert B.jac_from_mont_limbs(got.reshape(1, 12))[0] == exp


def test_c_eval_kate_batchinv():
    n = 777
    poly = random_scalars(n, 3)
    x = random_scalars(1, 4)[0]
    pm, xm = B.to_mont_limbs(poly), B.to_mont_limbs([x])[0]
    assert B.from_mont_limbs(OC.eval_polynomial(pm, xm))[0] == OP.eval_polynomial(poly, x)
    assert np.array_equal(OC.kate_division(pm, xm), B.to_mont_limbs(OP.kate_division(poly, x)))
    poly[5] = 0
    assert np.array_equal(OC.batch_invert(B.to_mont_limbs(poly)), B.to_mont_limbs(OP.batch_invert(poly)))


def test_c_g1_mul_and_affine():
    P = random_points(1, 9)[0]
    k = random_scalars(1, 10)[0]
    j = OC.g1_mul(B.points_to_mont_limbs([P])[0], B.to_mont_limbs([k])[0])
    assert B.points_from_mont_limbs(OC.g1_to_affine(j))[0] == B.g1_mul(P, k)
