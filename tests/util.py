"""Shared helpers for the parity tests (seeded synthetic inputs)."""
import numpy as np

from oracle import bn254 as B


def random_points(n, seed):
    """n pseudo-random G1 points: P_i = (a + i*b) * G built with additions only."""
    rng = B.Xoshiro256ss(seed)
    base = B.jac_mul(B.to_jac(B.G1_GEN), B.fr_random(rng))
    step = B.jac_mul(B.to_jac(B.G1_GEN), B.fr_random(rng))
    out = []
    cur = base
    for _ in range(n):
        out.append(cur)
        cur = B.jac_add(cur, step)
    return B.batch_to_affine(out)


def random_scalars(n, seed):
    rng = B.Xoshiro256ss(seed)
    return [B.fr_random(rng) for _ in range(n)]


def jac_limbs_to_affine(arr12):
    return B.jac_from_mont_limbs(np.asarray(arr12, dtype=np.uint64).reshape(1, 12))[0]
