import time, sys, numpy as np, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sha2_on_cq_halo2_amd import Context
from oracle import bn254 as B
from tests.util import random_points
ctx = Context(0)
base = B.points_to_mont_limbs(random_points(1 << 10, 3))
rs = np.random.RandomState(1)
for logn in [int(x) for x in sys.argv[1].split(',')]:
    n = 1 << logn
    pts = np.tile(base, (max(n >> 10, 1), 1))[:n]
    uni = rs.randint(0, 2**63, size=(n, 4), dtype=np.int64).astype(np.uint64); uni[:, 3] &= np.uint64((1 << 60) - 1)
    dpts = ctx.to_device(pts); dsc = ctx.to_device(uni)
    best = None
    for c in range(max(5, logn - 8), 16):
        ctx.set_msm_window(c)
        ctx.best_multiexp_dev(dsc, dpts, n)
        reps = 5 if logn < 20 else 2
        t = time.time()
        for _ in range(reps): ctx.best_multiexp_dev(dsc, dpts, n)
        dt = (time.time() - t) / reps
        print(f"n=2^{logn} c={c}: {dt*1e3:.3f} ms {n/dt/1e6:.1f} Mscalar/s", flush=True)
        if best is None or dt < best[1]: best = (c, dt)
    print(f"BEST n=2^{logn}: c={best[0]} {best[1]*1e3:.3f} ms", flush=True)
