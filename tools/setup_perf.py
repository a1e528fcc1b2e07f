"""Setup-path timings on the GPU: ParamsKZG::downsize / g_to_lagrange and CQ table preprocessing (reference O(N^2)
construction vs FK-style)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sha2_on_cq_halo2_amd import Context, ParamsKZG, StaticTable
from sha2_on_cq_halo2_amd.api import fr_to_mont
from sha2_on_cq_halo2_amd.sha_circuit import small_to_mont

ctx = Context(0)
s = fr_to_mont(0x1234567890ABCDEF1234567)
for k in (14, 18):
    p = ParamsKZG.setup_from_toxic_waste(ctx, k + 1, s)
    ctx.sync(); t = time.time(); d = p.downsize(k); ctx.sync()
    print("downsize 2^%d -> 2^%d (g_to_lagrange): %.1f ms" % (k + 1, k, (time.time() - t) * 1e3), flush=True)
for logn in (10, 12, 16):
    N = 1 << logn
    srs = ParamsKZG.setup_from_toxic_waste(ctx, logn, s).download()[0]
    vals = small_to_mont(np.arange(N))
    ctx.sync(); t = time.time(); fk = StaticTable.new_fk(ctx, vals, srs); ctx.sync(); tfk = time.time() - t
    line = "table N=2^%d: FK %.1f ms" % (logn, tfk * 1e3)
    if logn <= 12:
        t = time.time(); ref = StaticTable.new(ctx, vals, srs); ctx.sync(); tref = time.time() - t
        line += ", reference construction on the GPU %.1f ms, equal: %s" % (tref * 1e3, np.array_equal(fk.download_qs(), ref.download_qs()))
    print(line, flush=True)
ctx.close()
