"""Batched MSM timing: `batch` independent 2^k-point MSMs over ONE precomputed-table base array in a single launch.
Used with rocprofv3 --kernel-trace to compare per-entry kernel rates across batch sizes.
  python3 tools/msm_batch_perf.py 18 2,8,21"""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sha2_on_cq_halo2_amd import Context, ParamsKZG
from sha2_on_cq_halo2_amd.api import fr_to_mont

k = int(sys.argv[1]) if len(sys.argv) > 1 else 18
batches = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [2, 8, 21]
n = 1 << k
ctx = Context(0)
s = fr_to_mont(0x1234567890ABCDEF1234567890ABCDEF)
params = ParamsKZG.setup_from_toxic_waste(ctx, k, s)
rs = np.random.RandomState(1)
bufs = []
for j in range(max(batches)):
    a = rs.randint(0, 2**63, size=(n, 4), dtype=np.int64).astype(np.uint64)
    a[:, 3] &= np.uint64((1 << 60) - 1)
    bufs.append(ctx.to_device(a))
for b in batches:
    ptrs = [x.ptr for x in bufs[:b]]
    ctx.msm_batch_dev(ptrs, params.g_dev, n)
    t = time.time()
    for _ in range(3):
        ctx.msm_batch_dev(ptrs, params.g_dev, n)
    dt = (time.time() - t) / 3
    print("batch %2d x 2^%d: %.3f ms  %.1f Mscalar/s" % (b, k, dt * 1e3, b * n / dt / 1e6), flush=True)
