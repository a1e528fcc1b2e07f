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


@pytest.mark.parametrize("n", [300_001, 400_003, 3_200_001])  # 4, 8 and 16 elements per lane (poly_batch_invert's choice by size)
def test_batch_invert_large_arrays_every_lane_count(ctx, n):
    """Size-independent check at sizes the big-int oracle does not loop over: the words are a R mod p for some a, so the
    result words must be R^2 / w mod p -- compared on a sample of rows, the first and the last block included; zeros stay
    zero (ff::BatchInvert, arithmetic/curves' `batch_invert` call sites poly.rs:192,232)."""
    rs = np.random.RandomState(n & 0xFFFF)
    w = rs.randint(0, 2**63, size=(n, 4), dtype=np.int64).astype(np.uint64)
    w[:, 3] &= np.uint64((1 << 60) - 1)
    w[::7] = 0
    got = ctx.batch_invert(w.copy())
    assert got.shape == w.shape and not got[::7].any()
    rows = np.unique(np.concatenate([np.arange(0, 40), np.arange(n - 40, n), rs.randint(0, n, size=1500)]))
    r2 = pow(2, 512, B.R_MOD)
    for i in rows:
        x = sum(int(w[i, q]) << (64 * q) for q in range(4))
        y = sum(int(got[i, q]) << (64 * q) for q in range(4))
        assert y == (r2 * pow(x, -1, B.R_MOD) % B.R_MOD if x else 0), i
