"""GPU parity: EvaluationDomain transforms, eval_polynomial, kate_division, batch_invert vs the oracle."""
import numpy as np
import pytest

from oracle import bn254 as B
from oracle import poly as OP
from tests.util import random_scalars

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("j,k", [(3, 3), (3, 5), (2, 4), (4, 6), (9, 7), (3, 11), (5, 12)])
def test_domain_transforms(ctx, j, k):
    from sha2_on_cq_halo2_amd import EvaluationDomain

    od = OP.EvaluationDomain(j, k)
    gd = EvaluationDomain(ctx, j, k)
    assert gd.extended_k == od.extended_k
    cs = gd.constants()
    assert B.from_mont_limbs(cs["omega"])[0] == od.omega
    assert B.from_mont_limbs(cs["omega_inv"])[0] == od.omega_inv
    assert B.from_mont_limbs(cs["extended_omega"])[0] == od.extended_omega
    a = random_scalars(od.n, 100 * j + k)
    coeff = od.lagrange_to_coeff(a)
    assert np.array_equal(gd.lagrange_to_coeff(B.to_mont_limbs(a)), B.to_mont_limbs(coeff))
    ext = od.coeff_to_extended(coeff)
    assert np.array_equal(gd.coeff_to_extended(B.to_mont_limbs(coeff)), B.to_mont_limbs(ext))
    e = random_scalars(od.extended_len, 7 * j + k)
    back = od.extended_to_coeff(e)
    assert np.array_equal(gd.extended_to_coeff(B.to_mont_limbs(e)), B.to_mont_limbs(back))


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 1000, 4096, 4097, 70000])
def test_eval_polynomial(ctx, n):
    poly = random_scalars(n, n + 1)
    x = random_scalars(1, 9)[0]
    got = ctx.eval_polynomial(B.to_mont_limbs(poly) if n else np.zeros((0, 4), dtype=np.uint64), B.to_mont_limbs([x])[0])
    assert B.from_mont_limbs(got)[0] == OP.eval_polynomial(poly, x)


@pytest.mark.parametrize("n", [1, 2, 3, 64, 65, 129, 4096, 4100, 70000])
def test_kate_division(ctx, n):
    a = random_scalars(n, n + 5)
    z = random_scalars(1, 11)[0]
    got = ctx.kate_division(B.to_mont_limbs(a), B.to_mont_limbs([z])[0])
    exp = OP.kate_division(a, z)
    assert got.shape[0] == n - 1
    if n > 1:
        assert np.array_equal(got, B.to_mont_limbs(exp))


def test_kate_division_at_zero(ctx):
    a = random_scalars(300, 1)
    got = ctx.kate_division(B.to_mont_limbs(a), B.to_mont_limbs([0])[0])
    assert np.array_equal(got, B.to_mont_limbs(a[1:]))


@pytest.mark.parametrize("n", [0, 1, 7, 8, 9, 1000, 5000])
def test_batch_invert(ctx, n):
    a = random_scalars(n, n + 2)
    for i in range(0, n, 5):
        a[i] = 0
    got = ctx.batch_invert(B.to_mont_limbs(a) if n else np.zeros((0, 4), dtype=np.uint64))
    if n:
        assert np.array_equal(got, B.to_mont_limbs(OP.batch_invert(a)))
