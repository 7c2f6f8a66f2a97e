"""Print the per-kernel rows of a rocprofv3 kernel_stats CSV (name, calls, average ns, share)."""
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[: int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
        print(r["Name"][:90].ljust(90), r["Calls"], "%.1f us" % (float(r["AverageNs"]) / 1e3), r["Percentage"])
