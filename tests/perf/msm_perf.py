import os
import time, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from sha2_on_cq_halo2_amd import Context
from oracle import bn254 as B
from tests.util import random_points
ctx = Context(0)
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 18
cs = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [0, 11, 12, 13, 14, 15]
n = 1 << logn
base = random_points(1 << 12, 3)
pts = np.tile(B.points_to_mont_limbs(base), (n >> 12, 1))
rs = np.random.RandomState(1)
def mont(vals): return B.to_mont_limbs(vals)
uni = rs.randint(0, 2**63, size=(n, 4), dtype=np.int64).astype(np.uint64); uni[:, 3] &= np.uint64((1 << 60) - 1)
small = np.tile(mont([int(x) for x in rs.randint(0, 4096, size=4096)]), (n >> 12, 1))
bits = np.tile(mont([int(x) for x in rs.randint(0, 2, size=4096)]), (n >> 12, 1))
dpts = ctx.to_device(pts)
for name, sc in (("uniform", uni), ("limb12", small), ("bits", bits)):
    dsc = ctx.to_device(sc)
    for c in cs:
        ctx.set_msm_window(c)
        r = ctx.best_multiexp_dev(dsc, dpts, n)
        t = time.time(); reps = 5
        for _ in range(reps): r = ctx.best_multiexp_dev(dsc, dpts, n)
        dt = (time.time() - t) / reps
        print(f"{name} n=2^{logn} c={c}: {dt*1e3:.3f} ms  {n/dt/1e6:.1f} Mscalar/s", flush=True)
