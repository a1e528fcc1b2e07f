"""GPU parity: SRS construction (`ParamsKZG::setup_from_toxic_waste`) vs the oracle."""
import numpy as np
import pytest

from oracle import bn254 as B
from oracle import cbind as OC
from oracle import kzg
from tests.util import random_scalars

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k", [0, 1, 3, 6])
def test_params_setup_matches_oracle(ctx, k):
    from sha2_on_cq_halo2_amd import ParamsKZG

    s = B.fr_random(B.Xoshiro256ss(77 + k))
    op = kzg.ParamsKZG(k, s)
    p = ParamsKZG.setup_from_toxic_waste(ctx, k, B.to_mont_limbs([s])[0])
    g, gl = p.download()
    assert np.array_equal(g, B.points_to_mont_limbs(op.g))
    assert np.array_equal(gl, B.points_to_mont_limbs(op.g_lagrange))


def test_srs_pairing_free_consistency_k12(ctx):
    """Size-independent property on a larger SRS: commit(lagrange_to_coeff(a)) == commit_lagrange(a)
    (reference test `test_commit_lagrange`, poly/kzg/commitment.rs:570-593) with a = delta_3,
    whose coefficient form is known in closed form: coeff_j = w^(-3j)/n."""
    from sha2_on_cq_halo2_amd import ParamsKZG
    from oracle.poly import EvaluationDomain
    from tests.util import jac_limbs_to_affine

    k = 12
    n = 1 << k
    s = B.fr_random(B.Xoshiro256ss(5))
    p = ParamsKZG.setup_from_toxic_waste(ctx, k, B.to_mont_limbs([s])[0])
    dom = EvaluationDomain(2, k)
    ninv = pow(n, B.R_MOD - 2, B.R_MOD)
    w3inv = pow(dom.omega_inv, 3, B.R_MOD)
    coeffs = []
    cur = ninv
    for _ in range(n):
        coeffs.append(cur)
        cur = cur * w3inv % B.R_MOD
    evals = [0] * n
    evals[3] = 1
    c1 = jac_limbs_to_affine(p.commit(B.to_mont_limbs(coeffs)))
    c2 = jac_limbs_to_affine(p.commit_lagrange(B.to_mont_limbs(evals)))
    assert c1 == c2 and c1 is not None
    # and [L_3(s)]_1 itself
    _, gl = p.download()
    assert B.points_from_mont_limbs(gl[3:4])[0] == c1


@pytest.mark.parametrize("k_big,k_small", [(6, 4), (10, 9), (11, 11)])
def test_params_downsize_g_to_lagrange(ctx, k_big, k_small):
    """`ParamsKZG::downsize` (kzg/commitment.rs:480-492) -> `g_to_lagrange` (arithmetic.rs:277-301), an inverse FFT over
    G1 on the GPU: the downsized key equals a key set up directly at the smaller size from the same toxic waste
    (whose Lagrange basis comes from the closed form), point for point."""
    from sha2_on_cq_halo2_amd import ParamsKZG

    s = B.to_mont_limbs([B.fr_random(B.Xoshiro256ss(77))])[0]
    big = ParamsKZG.setup_from_toxic_waste(ctx, k_big, s)
    small = ParamsKZG.setup_from_toxic_waste(ctx, k_small, s)
    down = big.downsize(k_small)
    g0, gl0 = small.download()
    g1, gl1 = down.download()
    assert np.array_equal(g0, g1)
    assert np.array_equal(gl0, gl1)
    # and commitments through the downsized key (its window tables are built from the new arrays)
    a = B.to_mont_limbs(random_scalars(1 << k_small, 3))
    assert np.array_equal(OC.g1_to_affine(down.commit_lagrange(a)), OC.g1_to_affine(small.commit_lagrange(a)))
