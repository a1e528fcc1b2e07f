import time, sys, numpy as np, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sha2_on_cq_halo2_amd import Context
from oracle import bn254 as B
from tests.util import random_points
ctx = Context(0)
logn = int(sys.argv[1]); c = int(sys.argv[2]); kind = sys.argv[3] if len(sys.argv) > 3 else "uniform"
n = 1 << logn
base = random_points(1 << 12, 3)
pts = np.tile(B.points_to_mont_limbs(base), (n >> 12, 1))
rs = np.random.RandomState(1)
if kind == "uniform":
    sc = rs.randint(0, 2**63, size=(n, 4), dtype=np.int64).astype(np.uint64); sc[:, 3] &= np.uint64((1 << 60) - 1)
elif kind == "limb12":
    sc = np.tile(B.to_mont_limbs([int(x) for x in rs.randint(0, 4096, size=4096)]), (n >> 12, 1))
else:
    sc = np.tile(B.to_mont_limbs([int(x) for x in rs.randint(0, 2, size=4096)]), (n >> 12, 1))
dsc = ctx.to_device(sc); dpts = ctx.to_device(pts)
ctx.set_msm_window(c)
for _ in range(4):
    t=time.time(); r = ctx.best_multiexp_dev(dsc, dpts, n); print(time.time()-t, flush=True)
