"""Per-proof GPU timeline from a rocprofv3 --kernel-trace CSV: busy time, idle gaps and per-kernel sums inside the
last `window_ms` of the trace (one proof of tools/prove_large.py).

  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o run -- python3 tools/prove_large.py 18
  python3 tools/timeline.py gpurun_out/tl/*/run_kernel_trace.csv 15.0
"""
import csv
import sys
from collections import defaultdict

path, window_ms = sys.argv[1], float(sys.argv[2])
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
end = max(r[1] for r in rows)
lo = end - int(window_ms * 1e6)
win = [r for r in rows if r[0] >= lo]
busy = 0
cur_s, cur_e = win[0][0], win[0][1]
gaps = []
per = defaultdict(lambda: [0, 0])
prev_name = win[0][2]
for s, e, name in win:
    per[name][0] += e - s
    per[name][1] += 1
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, prev_name, name))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    prev_name = name
busy += cur_e - cur_s
span = end - win[0][0]
print("window %.2f ms: span %.2f ms, GPU busy %.2f ms, idle %.2f ms in %d gaps" % (window_ms, span / 1e6, busy / 1e6, (span - busy) / 1e6, len(gaps)))
print("-- kernels")
for name, (t, c) in sorted(per.items(), key=lambda kv: -kv[1][0]):
    print("  %-60s calls=%-4d total_us=%9.1f" % (name[:60], c, t / 1e3))
print("-- largest gaps (us): after -> before")
for g, a, b in sorted(gaps, reverse=True)[:25]:
    print("  %8.1f  %-45s -> %s" % (g / 1e3, a[:45], b[:45]))
small = sum(g for g, _, _ in gaps if g < 20000)
print("gaps < 20 us: %d totalling %.2f ms" % (sum(1 for g, _, _ in gaps if g < 20000), small / 1e6))
