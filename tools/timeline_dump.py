"""Chronological kernel list (with queue ids) of the last `window_ms` of a rocprofv3 kernel-trace CSV."""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("cq::", ""), r.get("Queue_Id", "?")))
rows.sort()
end = max(r[1] for r in rows)
lo = end - int(float(sys.argv[2]) * 1e6)
t0 = None
for s, e, n, q in rows:
    if s < lo:
        continue
    if t0 is None:
        t0 = s
    print("%8.1f %8.1f %7.1f  q%s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n[:40]))
