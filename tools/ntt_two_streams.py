"""Experiment: the same 2^k x batch transforms as tools/ntt_perf.py, but as `ways` contexts (streams) with batch / ways columns
each, launched from `ways` host threads -- do the passes of one stream fill the fill / drain phases of the other's?
   python3 tools/ntt_two_streams.py [k] [batch] [reps] [ways]"""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sha2_on_cq_halo2_amd import Context
from sha2_on_cq_halo2_amd.api import EvaluationDomain

k = int(sys.argv[1]) if len(sys.argv) > 1 else 18
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
ways = int(sys.argv[4]) if len(sys.argv) > 4 else 2
n = 1 << k
rs = np.random.RandomState(7)
per = batch // ways
ctxs, doms, srcs, dsts = [], [], [], []
for w in range(ways):
    ctx = Context(0)
    a = rs.randint(0, 2**63, size=(per * n, 4), dtype=np.int64).astype(np.uint64)
    a[:, 3] &= np.uint64((1 << 60) - 1)
    ctxs.append(ctx); doms.append(EvaluationDomain(ctx, 3, k)); srcs.append(ctx.to_device(a)); dsts.append(ctx.alloc(per * n * 32))

def run(w, r):
    c = ctxs[w]
    for _ in range(r):
        c._chk(c.lib.cq_lagrange_to_coeff_dev(doms[w].h, srcs[w].ptr, dsts[w].ptr, per))
    c.sync()

for w in range(ways): run(w, 2)
ts = [threading.Thread(target=run, args=(w, reps)) for w in range(ways)]
t = time.time()
for th in ts: th.start()
for th in ts: th.join()
dt = time.time() - t
print("NTT 2^%d x %d as %d streams x %d: %.3f ms per batch, %.2f Gelem/s" % (k, batch, ways, per, dt / reps * 1e3, batch * n * reps / dt / 1e9))
if hasattr(ctxs[0].lib, "cq_debug_ntt_trace"):  # -DCQ_NTT_TRACE build: log of the last `ways` x 3 transforms
    import ctypes
    lib = ctxs[0].lib
    lib.cq_debug_ntt_trace.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
    buf = np.zeros((65536, 4), dtype=np.uint64)
    lib.cq_debug_ntt_trace(buf.ctypes.data, 65536)
    ts = [threading.Thread(target=run, args=(w, 3)) for w in range(ways)]
    for th in ts: th.start()
    for th in ts: th.join()
    cnt = lib.cq_debug_ntt_trace(buf.ctypes.data, 65536)
    os.makedirs("gpurun_out", exist_ok=True)
    np.save("gpurun_out/ntt_wg_trace_%dways_k%d_b%d.npy" % (ways, k, batch), buf[:cnt])
    print("trace records", cnt)
