"""Every call of the kernels whose name contains <substring> in a rocprofv3 kernel_trace.csv: start (ms from the first
kernel), duration (us), grid size -- in launch order.   python3 tools/kernel_calls.py <dir> <substring> [<substring> ...]"""
import csv, glob, sys
d, subs = sys.argv[1], sys.argv[2:]
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    t0 = min(int(r["Start_Timestamp"]) for r in rows)
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    for r in rows:
        if any(s in r["Kernel_Name"] for s in subs):
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            print("%10.3f ms  %9.1f us  grid %-9s  %s" % ((s - t0) / 1e6, (e - s) / 1e3, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r["Kernel_Name"][:70]))
