"""Wall time of consecutive k=18 proofs, with an idle gap or a profiling read before some of them (what makes the first
timed step of bench.py slower than the rest?).   python3 tools/step_times.py"""
import gc, os, sys, time
import torch
torch.cuda.set_device(0)
torch.cuda.synchronize()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sha2_on_cq_halo2_amd import Context
from sha2_on_cq_halo2_amd.api import PROF_MSM_ACCUMULATE
from sha2_on_cq_halo2_amd.sha_circuit import ShaCqWorkload
ctx = Context(0)
wl = ShaCqWorkload(ctx, 18)
def run(n, pre=None, label=""):
    ts = []
    for i in range(n):
        if pre: pre(i)
        wl.fill_witness()
        t = time.perf_counter(); wl.prove(seed=2 + i); ts.append((time.perf_counter() - t) * 1e3)
    print(label, " ".join("%.1f" % t for t in ts), flush=True)
run(12, None, "plain       ")
run(8, lambda i: time.sleep(0.002) if i % 4 == 0 else None, "sleep 2ms/4  ")
run(8, lambda i: time.sleep(0.05) if i % 4 == 0 else None, "sleep 50ms/4 ")
run(8, lambda i: ctx.sync() if i % 4 == 0 else None, "ctx.sync/4   ")
ctx.profile_enable(True)
run(8, None, "profile on  ")
run(8, lambda i: ctx.profile_read(PROF_MSM_ACCUMULATE) if i % 4 == 0 else None, "prof read/4  ")
run(8, lambda i: torch.cuda.synchronize() if i % 4 == 0 else None, "torch sync/4 ")
run(8, lambda i: gc.collect() if i % 4 == 0 else None, "gc.collect/4 ")
x = torch.zeros(1 << 20, device="cuda")
run(8, lambda i: x.add_(1) if i % 4 == 0 else None, "torch kernel/4")
