"""Stand-alone timing of cq_batch_invert_dev (poly.hip batch_invert_kernel) on n elements, nothing else on the GPU.
   python3 tools/batch_invert_perf.py [n] [reps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sha2_on_cq_halo2_amd import Context

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4 * (1 << 18) + 4 * (1 << 16)
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ctx = Context(0)
rs = np.random.RandomState(3)
a = rs.randint(0, 2**63, size=(n, 4), dtype=np.int64).astype(np.uint64)
a[:, 3] &= np.uint64((1 << 60) - 1)
d = ctx.to_device(a)
ctx._chk(ctx.lib.cq_batch_invert_dev(ctx.h, d.ptr, n)); ctx.sync()
ts = []
for _ in range(reps):
    t = time.perf_counter()
    ctx._chk(ctx.lib.cq_batch_invert_dev(ctx.h, d.ptr, n)); ctx.sync()
    ts.append((time.perf_counter() - t) * 1e6)
ts.sort()
print("batch_invert of %d elements: median %.1f us, min %.1f us (wall, one call + sync)" % (n, ts[len(ts) // 2], ts[0]))
