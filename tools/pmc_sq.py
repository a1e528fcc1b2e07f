"""Summarise a rocprofv3 --pmc SQ_INSTS_VALU ... GRBM_GUI_ACTIVE pass: per launch of one kernel, the VALU wave-instructions,
GPU cycles and the VALU issue utilisation  SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs).
Usage: python tools/pmc_sq.py <pmc_dir> <kernel substring> <out.json> "<command that was profiled>" """
import collections, csv, glob, json, sys

d, kernel, out_path, command = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
rows = collections.OrderedDict()
for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if kernel not in r["Kernel_Name"]:
            continue
        key = int(r["Dispatch_Id"])
        rows.setdefault(key, {})[r["Counter_Name"]] = rows.get(key, {}).get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
launches = []
for i, (key, c) in enumerate(sorted(rows.items())):
    cycles = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    valu = c.get("SQ_INSTS_VALU", 0.0)
    launches.append({"launch": i, "valu_wave_instructions": valu, "gpu_cycles": cycles,
                     "valu_issue_utilisation": valu * 4.0 / (1024.0 * cycles) if cycles else None, "waves": c.get("SQ_WAVES")})
json.dump({"command": command,
           "note": "GRBM_GUI_ACTIVE is summed over the 8 XCDs (divided by 8 here); a wave64 VALU instruction occupies its SIMD "
                   "for 4 cycles; 1024 SIMDs.  Launches in dispatch order: 4 per proof since round 2 (advice + m | round 2 | h | W).",
           "kernels": {kernel: launches}}, open(out_path, "w"), indent=1)
for l in launches:
    print(l)
