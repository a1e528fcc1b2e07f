"""Per-kernel sums and gaps of exactly ONE proof, from the chronological dump tools/timeline_dump.py writes (a window of a fixed
length around the last proof of tools/prove_large.py usually holds the tail of the previous proof as well): the proof is taken
to start with the upload before its last `advice_fill_kernel`.
   python3 tools/timeline_dump.py <run_kernel_trace.csv> 8.6 > dump.txt;  python3 tools/timeline_one_proof.py dump.txt"""
import sys
from collections import defaultdict

rows = []
for ln in open(sys.argv[1]):
    p = ln.split()
    rows.append((float(p[0]), float(p[1]), float(p[2]), p[3], " ".join(p[4:])))
idx = max(i for i, r in enumerate(rows) if "advice_fill" in r[4])
sel = rows[max(idx - 1, 0):]
t0, end = sel[0][0], max(r[1] for r in sel)
per = defaultdict(lambda: [0.0, 0])
busy, cur_s, cur_e, gaps, prev = 0.0, sel[0][0], sel[0][1], [], sel[0][4]
for s, e, d, q, n in sel:
    per[n][0] += d
    per[n][1] += 1
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, prev, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
    prev = n
busy += cur_e - cur_s
print("one proof (from its first upload to its last copy): span %.2f ms, GPU busy %.2f ms, idle %.2f ms in %d gaps"
      % ((end - t0) / 1e3, busy / 1e3, (end - t0 - busy) / 1e3, len(gaps)))
print("-- kernels")
for n, (t, c) in sorted(per.items(), key=lambda kv: -kv[1][0]):
    print("  %-44s calls=%-3d total_us=%8.1f" % (n[:44], c, t))
print("-- largest gaps (us): after -> before")
for g, a, b in sorted(gaps, reverse=True)[:12]:
    print("  %7.1f  %-40s -> %s" % (g, a[:40], b[:40]))
