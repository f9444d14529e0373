"""print the per-kernel summary of a rocprofv3 --kernel-trace --stats --output-format csv run: python tools/kstats.py <dir>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    print("%-58s calls %6s avg %8.2f us min %6.2f max %7.2f %5.1f%%" % (r["Name"][:58], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
print("total %.3f ms  (%s)" % (tot / 1e6, f))
