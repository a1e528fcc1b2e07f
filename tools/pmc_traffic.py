"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes: per-kernel HBM-side bytes per launch.
Usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json>
FETCH_SIZE/WRITE_SIZE are in KiB (MI355X_MICROARCH.md, rocprofv3 section).  On gfx950 FETCH_SIZE reports
half the bytes of wide coalesced streaming reads; both the raw and the doubled figure are recorded."""
import csv, glob, json, sys, collections

def load(d, counter):
    rows = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                rows[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return rows

fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, []), write.get(k, [])
    name = k.split("(")[0]
    out[name] = {
        "launches": max(len(f), len(w)),
        "fetch_kib_per_launch_raw": sum(f) / len(f) if f else None,
        "write_kib_per_launch": sum(w) / len(w) if w else None,
    }
    if f and w:
        fr, wr = sum(f) / len(f) * 1024, sum(w) / len(w) * 1024
        out[name]["hbm_bytes_per_launch_raw"] = fr + wr
        out[name]["hbm_bytes_per_launch_fetch_x2"] = 2 * fr + wr
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in out.items():
    if v.get("hbm_bytes_per_launch_raw"):
        print("%-40s launches=%-4d raw=%.2f MB  fetchx2=%.2f MB" % (k[:40], v["launches"], v["hbm_bytes_per_launch_raw"] / 1e6, v["hbm_bytes_per_launch_fetch_x2"] / 1e6))
