"""Per-kernel summary of a rocprofv3 --pmc pass: python tools/pmc_summarize.py <counter_collection.csv> <COUNTER>
prints kernel, grid size, launches, median and max of the counter (summed over the XCDs' instances per dispatch)."""
import csv
import statistics
import sys
from collections import defaultdict

path, counter = sys.argv[1], sys.argv[2]
per_dispatch = defaultdict(float)
meta = {}
with open(path) as fh:
    for row in csv.DictReader(fh):
        if row.get("Counter_Name") != counter:
            continue
        key = row["Dispatch_Id"]
        per_dispatch[key] += float(row["Counter_Value"])
        name = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").replace("dcora::", "")
        meta[key] = (name.split("(")[0], row.get("Grid_Size", ""))
groups = defaultdict(list)
for k, v in per_dispatch.items():
    groups[meta[k]].append(v)
out = csv.writer(sys.stdout)
out.writerow(["kernel", "grid_size", "launches", counter + "_median", counter + "_max"])
for (name, grid), vals in sorted(groups.items()):
    out.writerow([name, grid, len(vals), "%.3f" % statistics.median(vals), "%.3f" % max(vals)])
