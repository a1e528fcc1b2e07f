import time, sys, numpy as np, ctypes as C
sys.path.insert(0, '.')
from sha2_on_cq_halo2_amd import Context
ctx = Context(0)
lanes = 256*256*8
buf = ctx.alloc(lanes*32)
for which in (0,1):
    for iters in (256, 4096):
        ctx._chk(ctx.lib.cq_bench_modmul_dev(ctx.h, buf.ptr, lanes, iters, which)); ctx.sync()
        t=time.time(); ctx._chk(ctx.lib.cq_bench_modmul_dev(ctx.h, buf.ptr, lanes, iters, which)); ctx.sync(); dt=time.time()-t
        print(f"modmul which={which} lanes={lanes} iters={iters}: {dt*1e3:.3f} ms -> {lanes*iters/dt/1e9:.2f} Gmul/s", flush=True)
