"""Stand-alone NTT timing: `reps` batched lagrange_to_coeff transforms of `batch` columns of 2^k (for kernel traces).
   python3 tools/ntt_perf.py [k] [batch] [reps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sha2_on_cq_halo2_amd import Context
from sha2_on_cq_halo2_amd.api import EvaluationDomain

k = int(sys.argv[1]) if len(sys.argv) > 1 else 18
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
ctx = Context(0)
dom = EvaluationDomain(ctx, 3, k)
n = 1 << k
rs = np.random.RandomState(7)
a = rs.randint(0, 2**63, size=(batch * n, 4), dtype=np.int64).astype(np.uint64)
a[:, 3] &= np.uint64((1 << 60) - 1)
src, dst = ctx.to_device(a), ctx.alloc(batch * n * 32)
ctx._chk(ctx.lib.cq_lagrange_to_coeff_dev(dom.h, src.ptr, dst.ptr, batch)); ctx.sync()
t = time.time()
for _ in range(reps):
    ctx._chk(ctx.lib.cq_lagrange_to_coeff_dev(dom.h, src.ptr, dst.ptr, batch))
ctx.sync()
dt = time.time() - t
print("NTT 2^%d x %d: %.3f ms per batch, %.2f Gelem/s" % (k, batch, dt / reps * 1e3, batch * n * reps / dt / 1e9))
