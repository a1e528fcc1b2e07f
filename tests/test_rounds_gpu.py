"""GPU parity, stage by stage: the stand-alone CQ entry points (cq_cq_round1_dev, cq_cq_round2_dev, cq_quotient_dev;
SURVEY.md 8(b) rows 5-6) against the intermediate values of the oracle's create_proof (`ProofTrace.points / polys`,
oracle/cq_prover.py) for the same key, witness, blinding rows and challenges."""
import numpy as np
import pytest

from oracle import bn254 as B

from .test_prover_gpu import _prove_both, _setup

pytestmark = pytest.mark.gpu


def _spread(x):
    r = 0
    for i in range(16):
        r |= ((x >> i) & 1) << (2 * i)
    return r


def _pts(points):
    return B.points_to_mont_limbs(list(points))


@pytest.mark.parametrize("k,nbits,pairs", [(5, 5, 1), (7, 6, 3)])
def test_cq_rounds_and_quotient_match_the_oracle_trace(ctx, k, nbits, pairs):
    N, n = 1 << nbits, 1 << k
    tv = {"dense": list(range(N)), "spread": [_spread(i) for i in range(N)]}
    lookups = [[(2 * p, "dense"), (2 * p + 1, "spread")] for p in range(pairs)]
    env = _setup(ctx, k, tv, lookups, 2 * pairs, 300 + k, srs_len=max(N, n))
    bf, u = 5, n - 6
    rng = B.Xoshiro256ss(k)
    advice = []
    for p in range(pairs):
        vals = [rng.next_u64() % N for _ in range(u - 2)]
        advice += [vals, [_spread(v) for v in vals]]
    seed = 4000 + k
    tr, proof = _prove_both(env, advice, seed)
    assert proof == tr.proof
    # the blinded columns create_proof commits: rows u.. are the first draws of the RNG stream (prover.rs:346-350)
    r2 = B.Xoshiro256ss(seed)
    cols = []
    for col in advice:
        full = list(col) + [0] * (u - len(col)) + [B.fr_random(r2) for _ in range(bf + 1)]
        cols.append(ctx.to_device(B.to_mont_limbs(full)))
    gpk = env["gpk"]
    L = pairs
    theta, beta, y = (B.to_mont_limbs([tr.challenges[c]])[0] for c in ("theta", "beta", "y"))

    # ---- round 1: f, m and their commitments
    f, m, cm1 = gpk.cq_round1([c.ptr for c in cols], theta)
    for l in range(L):
        assert np.array_equal(cm1[l], _pts([tr.points[f"f_cm{l}"], tr.points[f"m_cm{l}"]])), "round 1 commitments"
    mh = m.download((L, N), dtype=np.uint32)
    assert int(mh.sum()) == L * u and (mh[:, 0] >= 2).all()  # every usable row counted once; the unassigned rows hit index 0

    # ---- round 2: b, f in coefficient form, the five commitments, a(0)
    b, fc, cm2, a0 = gpk.cq_round2(f, m, theta, beta)
    bh, fh = b.download((L, n, 4)), fc.download((L, n, 4))
    for l in range(L):
        exp = _pts([tr.points[f"{name}{l}"] for name in ("a_cm", "qa_cm", "a0_cm", "b0_cm", "p_cm")])
        assert np.array_equal(cm2[l], exp), "round 2 commitments"
        assert np.array_equal(bh[l], B.to_mont_limbs(tr.polys[f"b{l}"]))
        assert np.array_equal(fh[l], B.to_mont_limbs(tr.polys[f"f{l}"]))
    # a(0) travels in the proof: the last scalar of each lookup's three (b_0(x), f(x), a(0))
    dom = env["opk"].domain
    for l in range(L):
        b_poly = tr.polys[f"b{l}"]
        exp_a0 = (b_poly[0] * n - (bf + 1) * pow(tr.challenges["beta"], B.R_MOD - 2, B.R_MOD)) * pow(N, B.R_MOD - 2, B.R_MOD) % B.R_MOD
        assert B.from_mont_limbs(a0[l:l + 1])[0] == exp_a0

    # ---- quotient: evaluate_h alone, then with the division by the vanishing polynomial
    ext = dom.extended_len
    h = gpk.cq_quotient(b, fc, y, beta, divide=False, ext=ext)
    assert np.array_equal(h.download((ext, 4)), B.to_mont_limbs(tr.polys["h_ext"]))
    hd = gpk.cq_quotient(b, fc, y, beta, divide=True, ext=ext)
    assert np.array_equal(hd.download((ext, 4)), B.to_mont_limbs(dom.divide_by_vanishing_poly(list(tr.polys["h_ext"]))))
    # folding onto terms computed elsewhere: h_in = a previous h gives h_in * y^L + the same terms
    h2 = gpk.cq_quotient(b, fc, y, beta, h_in=h, divide=False, ext=ext)
    yv = tr.challenges["y"]
    exp2 = [(a * pow(yv, L, B.R_MOD) + t) % B.R_MOD for a, t in zip(tr.polys["h_ext"], tr.polys["h_ext"])]
    assert np.array_equal(h2.download((ext, 4)), B.to_mont_limbs(exp2))


def test_cq_round1_reports_lookup_failures(ctx):
    from sha2_on_cq_halo2_amd import CqError

    tv = {"t": list(range(16))}
    env = _setup(ctx, 4, tv, [[(0, "t")]], 1, 9, srs_len=16)
    col = ctx.to_device(B.to_mont_limbs([3, 5, 99] + [0] * 13))
    with pytest.raises(CqError) as e:
        env["gpk"].cq_round1([col.ptr], B.to_mont_limbs([7])[0])
    assert e.value.code == -4
