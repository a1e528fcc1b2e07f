"""Long-running MSM parity stress (by hand on the GPU box): random lengths and scalar distributions, plain and
table mode, single and batched launches, against the C oracle's best_multiexp.
   python3 tests/perf/msm_stress.py [seconds] [seed]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import bn254 as B
from oracle import cbind as OC
from sha2_on_cq_halo2_amd import Context
from tests.util import random_points

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rs = np.random.RandomState(seed)
ctx = Context(0)
NMAX = 1 << 18
base = B.points_to_mont_limbs(random_points(1 << 11, 900 + seed))
pts = np.tile(base, (NMAX >> 11, 1))
# a few identity points and duplicates in the base array
pts[5] = 0
pts[7] = pts[6]
dpts = ctx.to_device(pts)
ctx.sync()
table_ready = False


def scalars(n, kind):
    if kind == "uniform":
        a = rs.randint(0, 2**63, size=(n, 4), dtype=np.int64).astype(np.uint64)
        a[:, 3] &= np.uint64((1 << 60) - 1)
        return a
    if kind == "small16":
        vals = rs.randint(0, 1 << 16, size=min(n, 4096))
    elif kind == "small32":
        vals = rs.randint(0, 1 << 32, size=min(n, 4096))
    elif kind == "bits":
        vals = rs.randint(0, 2, size=min(n, 4096))
    elif kind == "sparse":
        vals = rs.randint(0, 1 << 62, size=min(n, 4096)) * (rs.rand(min(n, 4096)) < 0.05)
    else:  # const
        vals = np.full(min(n, 4096), int(rs.randint(1, 1 << 62)), dtype=object)
    m = B.to_mont_limbs([int(v) for v in vals])
    reps = (n + len(m) - 1) // len(m)
    return np.tile(m, (reps, 1))[:n].copy()


t0 = time.time()
cases = 0
while time.time() - t0 < budget:
    n = int(rs.choice([1, 2, 3, 17, 255, 256, 257, 1000, 4096, 5000, 33000, 70001, 131072, 200000, 262144]))
    kind = str(rs.choice(["uniform", "small16", "small32", "bits", "sparse", "const"]))
    use_table = bool(rs.randint(0, 2))
    if use_table and not table_ready:
        ctx.msm_precompute(dpts.ptr, NMAX)
        table_ready = True
    if not use_table and table_ready:
        continue  # tables stay registered for the array: plain mode was exercised before the first table case
    batch = int(rs.choice([1, 1, 2, 5]))
    scs = [scalars(n, kind) for _ in range(batch)]
    dsc = [ctx.to_device(s) for s in scs]
    if batch == 1:
        got = [OC.g1_to_affine(ctx.best_multiexp_dev(dsc[0], dpts, n))]
    else:
        res = ctx.msm_batch_dev([d.ptr for d in dsc], dpts.ptr, n)
        got = [OC.g1_to_affine(r) for r in res]
    for g, s in zip(got, scs):
        exp = OC.g1_to_affine(OC.best_multiexp(s, pts[:n]))
        assert np.array_equal(g, exp), (n, kind, use_table, batch)
    cases += 1
print("msm stress: %d cases ok in %.0f s (seed %d, tables %s)" % (cases, time.time() - t0, seed, table_ready))
