"""Print a rocprofv3 kernel_stats.csv compactly: name, calls, avg us, %."""
import csv, glob, sys
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
        print("==", f)
        for r in csv.DictReader(open(f)):
            print("  %-44s calls=%-4s avg_us=%10.1f  %6.2f%%" % (r["Name"][:44], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
