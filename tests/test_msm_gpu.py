"""GPU parity: HIP Pippenger MSM (C ABI) vs the oracle's restatement of `best_multiexp`."""
import os

import numpy as np
import pytest

from oracle import bn254 as B
from oracle import poly as OP
from tests.util import jac_limbs_to_affine, random_points, random_scalars

pytestmark = pytest.mark.gpu


def _check(ctx, scalars, points):
    exp = B.jac_to_affine(OP.best_multiexp(scalars, points))
    got = ctx.best_multiexp(B.to_mont_limbs(scalars), B.points_to_mont_limbs(points))
    assert jac_limbs_to_affine(got) == exp


@pytest.mark.parametrize("n", [0, 1, 2, 3, 4, 5, 31, 32, 33, 100, 257, 1000, 4096])
def test_msm_random(ctx, n):
    _check(ctx, random_scalars(n, 10 + n), random_points(n, 20 + n))


@pytest.mark.parametrize("c", [2, 3, 5, 8, 11, 13, 15])
def test_msm_window_sizes(ctx, c):
    n = 700
    ctx.set_msm_window(c)
    try:
        _check(ctx, random_scalars(n, 77), random_points(n, 78))
    finally:
        ctx.set_msm_window(0)


def test_msm_edge_scalars(ctx):
    n = 300
    pts = random_points(n, 5)
    sc = random_scalars(n, 6)
    sc[0] = 0
    sc[1] = 1
    sc[2] = B.R_MOD - 1
    sc[3] = B.R_MOD - 2
    sc[4] = (1 << 253) + 12345
    sc[5] = (1 << 128) - 1
    for i in range(6, 60):
        sc[i] = i - 6  # small scalars, many zero high windows
    _check(ctx, sc, pts)


def test_msm_identity_and_repeated_bases(ctx):
    n = 200
    pts = random_points(n, 9)
    for i in range(0, n, 7):
        pts[i] = None  # identity bases (0,0)
    for i in range(50, 120):
        pts[i] = pts[50]  # repeated base: forces the doubling branch inside buckets
    pts[130] = B.g1_neg(pts[131])  # P and -P
    sc = random_scalars(n, 11)
    for i in range(50, 120):
        sc[i] = 5  # same bucket
    sc[130] = sc[131] = 9
    _check(ctx, sc, pts)


def test_msm_all_same_point_small_scalars(ctx):
    n = 512
    pts = [B.G1_GEN] * n
    sc = [(i % 3) for i in range(n)]
    _check(ctx, sc, pts)


def test_msm_len_mismatch(ctx):
    from sha2_on_cq_halo2_amd import CqError

    with pytest.raises(CqError):
        ctx.best_multiexp(np.zeros((3, 4), dtype=np.uint64), np.zeros((2, 8), dtype=np.uint64))


def test_commit_equals_commit_lagrange(ctx):
    """Reference identity test `test_commit_lagrange` (poly/kzg/commitment.rs:570-593):
    commit(lagrange_to_coeff(a)) == commit_lagrange(a), on a true SRS."""
    from oracle import kzg
    from sha2_on_cq_halo2_amd import ParamsKZG

    k = 6
    s = B.fr_random(B.Xoshiro256ss(1234))
    op = kzg.ParamsKZG(k, s)
    params = ParamsKZG(ctx, k, B.points_to_mont_limbs(op.g), B.points_to_mont_limbs(op.g_lagrange))
    a = random_scalars(1 << k, 99)
    dom = OP.EvaluationDomain(2, k)
    coeffs = dom.lagrange_to_coeff(a)
    c1 = jac_limbs_to_affine(params.commit(B.to_mont_limbs(coeffs)))
    c2 = jac_limbs_to_affine(params.commit_lagrange(B.to_mont_limbs(a)))
    assert c1 == c2
    assert c1 == B.jac_to_affine(op.commit(coeffs))


@pytest.mark.parametrize("kind", ["bits", "limb12", "const"])
def test_msm_skewed_scalars_multilevel(ctx, kind):
    """Skewed digit distributions force the multi-level (bounded sub-list) accumulation path:
    0/1 selector-like columns, 12-bit SHA limbs, and one repeated full-size scalar."""
    n = 6000
    pts = random_points(n, 41)
    rng = B.Xoshiro256ss(42)
    if kind == "bits":
        sc = [rng.next_u64() & 1 for _ in range(n)]
    elif kind == "limb12":
        sc = [rng.next_u64() & 0xFFF for _ in range(n)]
    else:
        v = B.fr_random(rng)
        sc = [v] * n
    ctx.set_msm_window(9)
    try:
        _check(ctx, sc, pts)
    finally:
        ctx.set_msm_window(0)
    _check(ctx, sc, pts)


@pytest.mark.parametrize("n,kind", [(700, "uniform"), (5000, "uniform"), (5000, "bits"), (40000, "uniform"),
                                    (70000, "bits"), (70000, "const"), (33000, "limb12")])
def test_msm_precomputed_tables_same_result(ctx, n, kind):
    """Fixed-base mode (per-window tables, one bucket set) returns the same point as the plain
    pipeline and as the oracle; also on a prefix of the registered array."""
    from oracle import cbind as OC

    pts = B.points_to_mont_limbs(random_points(min(n, 2048), 91))
    pts = np.tile(pts, ((n + 2047) // 2048, 1))[:n]
    rng = B.Xoshiro256ss(92)
    if kind == "bits":  # one bucket (and one partition of the table-mode sort) takes every entry: many tiles, long lists
        sc = B.to_mont_limbs([rng.next_u64() & 1 for _ in range(n)])
    elif kind == "const":  # 17 buckets with n entries each
        sc = np.tile(B.to_mont_limbs([B.fr_random(rng)]), (n, 1))
    elif kind == "limb12":
        sc = np.tile(B.to_mont_limbs([rng.next_u64() & 0xFFF for _ in range(1000)]), (n // 1000, 1))
    else:
        rs = np.random.RandomState(n)
        sc = rs.randint(0, 2**63, size=(n, 4), dtype=np.int64).astype(np.uint64)
        sc[:, 3] &= np.uint64((1 << 60) - 1)
    dsc, dpts = ctx.to_device(sc), ctx.to_device(pts)
    plain = OC.g1_to_affine(ctx.best_multiexp_dev(dsc, dpts, n))
    exp = OC.g1_to_affine(OC.best_multiexp(sc, pts))
    assert np.array_equal(plain, exp)
    ctx.msm_precompute(dpts.ptr, n)
    pre = OC.g1_to_affine(ctx.best_multiexp_dev(dsc, dpts, n))
    assert np.array_equal(pre, exp)
    m = n - 7
    assert np.array_equal(OC.g1_to_affine(ctx.best_multiexp_dev(dsc, dpts, m)), OC.g1_to_affine(OC.best_multiexp(sc[:m], pts[:m])))


def test_msm_batch_with_repeated_scalar_vector(ctx):
    """MSMs of one launch over the same scalar vector share their sorted entry lists (b0 and p of a CQ lookup): the
    later one is not sorted again.  [s0, s1, s0] in one launch, table mode, against the C oracle."""
    from oracle import cbind as OC

    n = 40000
    pts = B.points_to_mont_limbs(random_points(2048, 77))
    pts = np.tile(pts, ((n + 2047) // 2048, 1))[:n]
    rs = np.random.RandomState(3)
    scs = []
    for _ in range(2):
        a = rs.randint(0, 2**63, size=(n, 4), dtype=np.int64).astype(np.uint64)
        a[:, 3] &= np.uint64((1 << 60) - 1)
        scs.append(a)
    dpts = ctx.to_device(pts)
    ctx.msm_precompute(dpts.ptr, n)
    d0, d1 = ctx.to_device(scs[0]), ctx.to_device(scs[1])
    res = ctx.msm_batch_dev([d0.ptr, d1.ptr, d0.ptr], dpts.ptr, n)
    exp = [OC.g1_to_affine(OC.best_multiexp(s, pts)) for s in scs]
    got = [OC.g1_to_affine(r) for r in res]
    assert np.array_equal(got[0], exp[0]) and np.array_equal(got[1], exp[1]) and np.array_equal(got[2], exp[0])


@pytest.mark.parametrize("c", [8, 12, 14, 16, 17, 18, 19, 20])
def test_msm_table_window_widths(ctx, c):
    """Tables of other window widths than the default 15 (cq_msm_set_table_window): narrower ones have fewer partitions,
    wider ones 2^(c-15) bucket sets per MSM and a refinement pass in the sort.  A launch of four MSMs (uniform scalars,
    a 0/1 column, one repeated value, and the first vector again) over 70 001 points against the C oracle."""
    from oracle import cbind as OC

    n = 70001
    pts = B.points_to_mont_limbs(random_points(2048, 300 + c))
    pts = np.tile(pts, ((n + 2047) // 2048, 1))[:n]
    rs = np.random.RandomState(c)
    uni = rs.randint(0, 2**63, size=(n, 4), dtype=np.int64).astype(np.uint64)
    uni[:, 3] &= np.uint64((1 << 60) - 1)
    rng = B.Xoshiro256ss(c)
    bits = B.to_mont_limbs([rng.next_u64() & 1 for _ in range(n)])
    const = np.tile(B.to_mont_limbs([B.fr_random(rng)]), (n, 1))
    vecs = [uni, bits, const]
    dpts = ctx.to_device(pts)
    ctx.set_msm_table_window(c)
    try:
        ctx.msm_precompute(dpts.ptr, n)
    finally:
        ctx.set_msm_table_window(0)
    dev = [ctx.to_device(v) for v in vecs]
    order = [0, 1, 2, 0]
    res = ctx.msm_batch_dev([dev[i].ptr for i in order], dpts.ptr, n)
    exp = [OC.g1_to_affine(OC.best_multiexp(v, pts)) for v in vecs]
    for j, i in enumerate(order):
        assert np.array_equal(OC.g1_to_affine(res[j]), exp[i]), "MSM %d differs from the oracle at c = %d" % (j, c)
    m = n - 4099  # a prefix of the registered array
    assert np.array_equal(OC.g1_to_affine(ctx.best_multiexp_dev(dev[0], dpts, m)), OC.g1_to_affine(OC.best_multiexp(uni[:m], pts[:m])))


@pytest.mark.parametrize("seed", range(int(os.environ.get("CQ_FUZZ_SEEDS", "8"))))  # CQ_FUZZ_SEEDS=100 for a long run
def test_msm_fuzz_table_mode(ctx, seed):
    """Fuzz of the table-mode launch: random length, table window width (8..20: every sort shape), number of MSMs, scalar
    mixes (uniform, small, 0/1, constant, sparse, zero) with repeated vectors and prefix lengths, against the C oracle."""
    from oracle import cbind as OC

    rs = np.random.RandomState(1000 + seed)
    n = int(rs.choice([rs.randint(1, 300), rs.randint(300, 5000), rs.randint(5000, 120000)]))
    c = int(rs.randint(8, 21))
    pts = B.points_to_mont_limbs(random_points(min(n, 512), 500 + seed))
    pts = np.tile(pts, ((n + 511) // 512, 1))[:n]
    if n > 3:
        pts[rs.randint(0, n)] = 0  # an identity base

    def vec(kind):
        a = np.zeros((n, 4), dtype=np.uint64)
        if kind == 0:
            a = rs.randint(0, 2**63, size=(n, 4), dtype=np.int64).astype(np.uint64)
            a[:, 3] &= np.uint64((1 << 60) - 1)
        elif kind == 1:
            a[:, 0] = rs.randint(0, 1 << int(rs.randint(1, 40)), size=n).astype(np.uint64)
        elif kind == 2:
            a = np.tile(B.to_mont_limbs([int(v) for v in rs.randint(0, 2, size=min(n, 64))]), ((n + 63) // 64, 1))[:n]
        elif kind == 3:
            a = np.tile(B.to_mont_limbs([B.fr_random(B.Xoshiro256ss(seed))]), (n, 1))
        elif kind == 4:
            u = rs.randint(0, 2**63, size=(n, 4), dtype=np.int64).astype(np.uint64)
            u[:, 3] &= np.uint64((1 << 60) - 1)
            step = int(rs.randint(2, 50))
            a[::step] = u[::step]
        return np.ascontiguousarray(a)

    count = int(rs.randint(1, 6))
    kinds = [int(k) for k in rs.randint(0, 6, size=count)]
    vecs = [vec(k) for k in kinds]
    order = list(range(count))
    if count > 1 and rs.randint(0, 2):
        order.append(int(rs.randint(0, count)))  # one vector twice: shared entry lists
    dpts = ctx.to_device(pts)
    ctx.set_msm_table_window(c)
    try:
        ctx.msm_precompute(dpts.ptr, n)
    finally:
        ctx.set_msm_table_window(0)
    dev = [ctx.to_device(v) for v in vecs]
    res = ctx.msm_batch_dev([dev[i].ptr for i in order], dpts.ptr, n)
    exp = [OC.g1_to_affine(OC.best_multiexp(v, pts)) for v in vecs]
    for j, i in enumerate(order):
        assert np.array_equal(OC.g1_to_affine(res[j]), exp[i]), "seed %d: MSM %d (kind %d) at c = %d, n = %d" % (seed, j, kinds[i], c, n)
    m = int(rs.randint(1, n + 1))
    assert np.array_equal(OC.g1_to_affine(ctx.best_multiexp_dev(dev[0], dpts, m)), OC.g1_to_affine(OC.best_multiexp(vecs[0][:m], pts[:m])))


def test_msm_table_window_argument_range(ctx):
    from sha2_on_cq_halo2_amd import CqError

    for bad in (1, 7, 21, 64):
        with pytest.raises(CqError):
            ctx.set_msm_table_window(bad)
    ctx.set_msm_table_window(0)


def test_msm_k18_round2_shaped_launch_matches_c_oracle(ctx):
    """One table-mode launch of the size a k = 18 proof really issues: 11 multiexps of 2^18 terms over a registered
    SRS array in ONE cq_msm_batch_dev call (~49 M sorted entries, equal sub-lists, shared lists for the repeated
    scalar vector), each compared with the C restatement of `best_multiexp` (arithmetic.rs:132-159).  Scalar mixes
    as in a proof: uniform field elements, 12-bit limbs, a 0/1 selector, one constant column, a sparse column."""
    from oracle import cbind as OC
    from sha2_on_cq_halo2_amd import ParamsKZG

    k = 18
    n = 1 << k
    s = B.to_mont_limbs([B.fr_random(B.Xoshiro256ss(0x18))])[0]
    params = ParamsKZG.setup_from_toxic_waste(ctx, k, s)  # g_lagrange registered: 17 window tables
    _, gl = params.download()
    rs = np.random.RandomState(18)

    def uniform():
        a = rs.randint(0, 2**63, size=(n, 4), dtype=np.int64).astype(np.uint64)
        a[:, 3] &= np.uint64((1 << 60) - 1)
        return a

    def small(bits):
        a = np.zeros((n, 4), dtype=np.uint64)
        a[:, 0] = rs.randint(0, 1 << bits, size=n).astype(np.uint64)
        return a  # raw limbs: x * 2^-256 as field elements, i.e. still "random" values but with a fixed pattern

    lim = np.tile(B.to_mont_limbs([int(v) for v in rs.randint(0, 4096, size=4096)]), (n // 4096, 1))
    bits = np.tile(B.to_mont_limbs([int(v) for v in rs.randint(0, 2, size=4096)]), (n // 4096, 1))
    const = np.tile(B.to_mont_limbs([B.fr_random(B.Xoshiro256ss(5))]), (n, 1))
    sparse = np.zeros((n, 4), dtype=np.uint64)
    sparse[::997] = uniform()[::997]
    vecs = [uniform(), uniform(), lim, bits, const, sparse, uniform(), small(40), uniform(), uniform()]
    dev = [ctx.to_device(v) for v in vecs]
    order = list(range(len(vecs))) + [0]  # 11 MSMs, the last over the first one's scalars again
    res = ctx.msm_batch_dev([dev[i].ptr for i in order], params.g_lagrange_dev, n)
    exp = [OC.g1_to_affine(OC.best_multiexp(v, gl)) for v in vecs]
    for j, i in enumerate(order):
        assert np.array_equal(OC.g1_to_affine(res[j]), exp[i]), "MSM %d of the launch differs from the oracle" % j
    params.close()


def test_table_width_does_not_depend_on_construction_order():
    """The window width of an array's tables comes from the array (and a key brings its small arrays to the width of its SRS
    tables), not from whatever was registered first on the context: a table config built under another width, then params
    and a key built normally, give the tables of a fresh context -- and the same proof bytes in every mix."""
    import types

    from sha2_on_cq_halo2_amd import Context, ParamsKZG
    from sha2_on_cq_halo2_amd.api import fr_to_mont
    from sha2_on_cq_halo2_amd.sha_circuit import ShaCqWorkload

    k, n = 11, 1 << 11
    c0 = Context(0)
    base = ShaCqWorkload(c0, k, pairs=2)
    p0 = base.prove(seed=3)
    assert c0.msm_table_width(base.params.g_dev, n) == 15
    c0.close()

    c1 = Context(0)
    c1.set_msm_table_window(12)
    narrow = ShaCqWorkload(c1, k, pairs=2)  # every table of this one: 12 bits
    assert c1.msm_table_width(narrow.params.g_dev, n) == 12
    assert narrow.prove(seed=3) == p0
    c1.set_msm_table_window(0)
    seed = 0x5348413243515F
    params = ParamsKZG.setup_from_toxic_waste(c1, k, fr_to_mont(seed * 0x9E3779B97F4A7C15 + 12345))
    assert c1.msm_table_width(params.g_dev, n) == 15  # its own length decides, not the 12-bit tables built before it
    mixed = ShaCqWorkload(c1, k, pairs=2, share=types.SimpleNamespace(k=k, params=params, cfg=narrow.cfg, dense=narrow.dense, spread=narrow.spread))
    assert mixed.prove(seed=3) == p0      # table SRS built at 12 bits, SRS at 15: the key added 15-bit tables for the small arrays
    assert narrow.prove(seed=3) == p0     # ... next to the 12-bit ones the first key still uses
    mixed.pk.close()
    assert narrow.prove(seed=3) == p0
    c1.close()


def test_sharded_key_leaves_the_tables_of_shared_srs_objects_alone():
    """cq_pk_set_sharding builds window tables for the rank's slices; the whole-array tables of params / table config are
    given up for that only while the key is their one user, and come back with world = 1 or when the key is destroyed."""
    import ctypes as C

    from sha2_on_cq_halo2_amd import Context
    from sha2_on_cq_halo2_amd.api import _ALLGATHER_T
    from sha2_on_cq_halo2_amd.sha_circuit import ShaCqWorkload

    k, n = 11, 1 << 11
    c = Context(0)
    a = ShaCqWorkload(c, k, pairs=2)
    b = ShaCqWorkload(c, k, pairs=2, share=a)
    p0 = a.prove(seed=4)
    never = _ALLGATHER_T(lambda *_: -1)  # no proof is made with the sharded key here: the registry is what is checked

    def shard(wl, rank, world):
        c._chk(c.lib.cq_pk_set_sharding(wl.pk.h, rank, world, C.cast(never, C.c_void_p) if world > 1 else None, None))

    g = a.params.g_dev
    shard(b, 1, 2)
    assert c.msm_table_width(g, n) == 15            # `a` shares the params: its whole-array tables stay
    assert a.prove(seed=4) == p0
    b.pk.close()                                    # destroying the sharded key takes nothing of a's with it
    assert c.msm_table_width(g, n) == 15 and a.prove(seed=4) == p0
    shard(a, 1, 2)                                  # now the only user: slices only
    lo = n // 2
    assert c.msm_table_width(g, n) == 0 and c.msm_table_width(g + 64 * lo, n - lo) == 15
    shard(a, 0, 1)
    assert c.msm_table_width(g, n) == 15 and a.prove(seed=4) == p0
    shard(a, 0, 2)
    a.pk.close()                                    # a sharded key that goes away restores what it had dropped
    assert c.msm_table_width(g, n) == 15
    c.close()


def test_srs_objects_destroyed_before_their_key():
    """The documented order is keys first, but a garbage collector (or a careless caller) may destroy the params / table
    config while a proving key still refers to them: they are then only marked released, the key keeps proving the same
    bytes, and the last key frees them -- no freed memory is touched in either order."""
    from sha2_on_cq_halo2_amd import Context
    from sha2_on_cq_halo2_amd.sha_circuit import ShaCqWorkload

    c = Context(0)
    a = ShaCqWorkload(c, 10, pairs=2)
    b = ShaCqWorkload(c, 10, pairs=2, share=a)
    p0 = a.prove(seed=8)
    a.params.close()  # while two keys use them
    a.cfg.close()
    assert a.prove(seed=8) == p0 and b.prove(seed=8) == p0
    a.pk.close()
    assert b.prove(seed=8) == p0
    b.pk.close()      # the last user: frees the released params and table config
    fresh = ShaCqWorkload(c, 10, pairs=2)
    assert fresh.prove(seed=8) == p0
    c.close()


def test_quad_add_matches_lane_serial_addition(tmp_path):
    """The four-lane addition of the launch tails (curve29.hpp quad_add; msm_rowcol_quad_kernel, msm_weighted_quad_kernel)
    against xyzz29_add on every kind of operand pair: general, either or both the identity, equal points (doubling),
    opposite points (cancellation).  Built from tools/micro/quad_add_test.hip with hipcc on the GPU box."""
    import shutil
    import subprocess

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "quad_add_test")
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-value", "-I", os.path.join(root, "include"),
                        "-I", os.path.join(root, "sha2_on_cq_halo2_amd", "csrc"), os.path.join(root, "tools", "micro", "quad_add_test.hip"),
                        "-o", exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr[-1000:]
    assert r.stdout.count("0 of 64 lanes disagree") == 6, r.stdout


def test_combine_wave_half_strides_over_more_slots_than_waves():
    """The combine levels' wave half is a capped grid whose waves stride over the sub-list slots (msm.hip: one block per four
    POSSIBLE slots was a million empty blocks per level at k = 22).  With the default cap of 2048 blocks only launches of
    2^20 equal scalars and more have more slots than waves, so a child process shrinks the cap to ONE block
    (CQ_MSM_COMBINE_WAVE_BLOCKS=1: four waves) and sums columns whose entries all fall into one bucket per window -- a
    constant column, a 0/1 column, a column of two values -- next to a uniform one: hundreds of slots per level, each wave
    takes dozens of them.  Compared with the C restatement of `best_multiexp` (arithmetic.rs:132-159)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import os, sys
import numpy as np
sys.path.insert(0, %r)
from oracle import bn254 as B
from oracle import cbind as OC
from tests.util import random_points
from sha2_on_cq_halo2_amd import Context
ctx = Context(0)
n = 1 << 15
rs = np.random.RandomState(77)
pts = np.tile(B.points_to_mont_limbs(random_points(512, 577)), (n // 512, 1))
one = lambda seed: np.tile(B.to_mont_limbs([B.fr_random(B.Xoshiro256ss(seed))]), (n, 1))
const = one(9)
bits = B.to_mont_limbs([int(v) for v in rs.randint(0, 2, size=n)])
two = np.where(rs.randint(0, 2, size=(n, 1)).astype(bool), const, one(10))
uni = rs.randint(0, 2**63, size=(n, 4), dtype=np.int64).astype(np.uint64); uni[:, 3] &= np.uint64((1 << 60) - 1)
vecs = [np.ascontiguousarray(v) for v in (const, bits, two, uni)]
dpts = ctx.to_device(pts)
ctx.msm_precompute(dpts.ptr, n)
dev = [ctx.to_device(v) for v in vecs]
res = ctx.msm_batch_dev([d.ptr for d in dev], dpts.ptr, n)
for j, v in enumerate(vecs):
    assert np.array_equal(OC.g1_to_affine(res[j]), OC.g1_to_affine(OC.best_multiexp(v, pts))), "MSM %%d differs" %% j
ctx2 = Context(0)  # and without tables (the caller's array, converted on the fly)
d2 = ctx2.to_device(pts)
for j, v in enumerate(vecs[:3]):
    got = ctx2.best_multiexp_dev(ctx2.to_device(v), d2, n)
    assert np.array_equal(OC.g1_to_affine(got), OC.g1_to_affine(OC.best_multiexp(v, pts))), "plain MSM %%d differs" %% j
print("ok")
""" % root
    env = dict(os.environ, CQ_MSM_COMBINE_WAVE_BLOCKS="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-3000:]
