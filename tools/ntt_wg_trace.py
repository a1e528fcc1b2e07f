"""Where and when the workgroups of the NTT pass kernel run (needs a build with -DCQ_NTT_TRACE: tools/ntt_wg_trace.sh).
Every workgroup logs wall_clock64() (100 MHz) at entry and after its last store, its XCC / SE / CU and its block index; this
prints, per pass launch: span, workgroup lifetimes, how long the first generation takes to be placed, the gap between a
workgroup leaving a CU and its successor starting there, and the occupancy over time.
   python3 tools/ntt_wg_trace.py [k] [batch]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sha2_on_cq_halo2_amd import Context
from sha2_on_cq_halo2_amd.api import EvaluationDomain

k = int(sys.argv[1]) if len(sys.argv) > 1 else 18
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ctx = Context(0)
dom = EvaluationDomain(ctx, 3, k)
n = 1 << k
rs = np.random.RandomState(7)
a = rs.randint(0, 2**63, size=(batch * n, 4), dtype=np.int64).astype(np.uint64)
a[:, 3] &= np.uint64((1 << 60) - 1)
src, dst = ctx.to_device(a), ctx.alloc(batch * n * 32)
lib = ctx.lib
lib.cq_debug_ntt_trace.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
buf = np.zeros((65536, 4), dtype=np.uint64)
for _ in range(3):
    ctx._chk(lib.cq_lagrange_to_coeff_dev(dom.h, src.ptr, dst.ptr, batch)); ctx.sync()
    cnt = lib.cq_debug_ntt_trace(buf.ctypes.data, 65536)
rec = buf[:cnt].copy()
os.makedirs("gpurun_out", exist_ok=True)
np.save("gpurun_out/ntt_wg_trace_k%d_b%d.npy" % (k, batch), rec)
t0, t1, hw, blk = rec[:, 0].astype(np.int64), rec[:, 1].astype(np.int64), rec[:, 2], rec[:, 3]
xcc = (hw >> np.uint64(32)).astype(np.int64) & 15
hwid = (hw & np.uint64(0xffffffff)).astype(np.int64)
cu = (hwid >> 8) & 15; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 7
place = ((xcc * 8 + se) * 2 + sh) * 16 + cu
print("records", cnt, "distinct CUs", len(set(place.tolist())), "XCCs", sorted(set(xcc.tolist())))
# split into launches by time gaps: sort by start
order = np.argsort(t0)
t0, t1, place, blk = t0[order], t1[order], place[order], blk[order]
launch_of = np.zeros(cnt, dtype=np.int64)
per = cnt // 3 if cnt % 3 == 0 else None
if per:
    launch_of = np.arange(cnt) // per
for L in sorted(set(launch_of.tolist())):
    m = launch_of == L
    s, e, pl = t0[m], t1[m], place[m]
    base = s.min()
    s = (s - base) / 100.0; e = (e - base) / 100.0  # microseconds
    life = e - s
    print("launch %d: %d workgroups, span %.1f us; lifetime mean %.1f min %.1f max %.1f us" % (L, m.sum(), e.max(), life.mean(), life.min(), life.max()))
    ss = np.sort(s)
    q = [ss[int(len(ss) * f) - 1] for f in (0.125, 0.25, 0.5)]
    print("   start time of the 1/8, 1/4, 1/2-th workgroup: %.1f %.1f %.1f us; last start %.1f us" % (q[0], q[1], q[2], ss[-1]))
    # per CU: workgroups resident over time, successor gaps
    gaps = []
    for c in set(pl.tolist()):
        idx = np.where(pl == c)[0]
        ends = np.sort(e[idx]); starts = np.sort(s[idx])
        # successor gap: for each start after the first 4, time since the earliest unfilled end
        later = starts[4:] if len(starts) > 4 else []
        for j, st in enumerate(later):
            if j < len(ends): gaps.append(st - ends[j])
    if gaps:
        gaps = np.array(gaps)
        print("   successor start - predecessor end on the same CU: mean %.2f median %.2f max %.2f us (%d pairs)" % (gaps.mean(), np.median(gaps), gaps.max(), len(gaps)))
    per_cu = np.bincount(pl.astype(np.int64))
    per_cu = per_cu[per_cu > 0]
    print("   workgroups per CU: min %d max %d" % (per_cu.min(), per_cu.max()))
    # occupancy over time in 5 us bins
    T = e.max(); bins = np.arange(0, T + 5, 5.0)
    occ = [(np.minimum(e, b + 5) - np.maximum(s, b)).clip(min=0).sum() / 5.0 for b in bins[:-1]]
    print("   resident workgroups per 5 us bin:", " ".join("%d" % o for o in occ))
