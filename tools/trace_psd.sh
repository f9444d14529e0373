#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/psdtr
rm -rf $o; mkdir -p $o
rocprofv3 --kernel-trace --stats --output-format csv -d $o -o t -- python3 tools/prof_psd.py ${1:-lattice} > $o/run.log 2>&1
cat $o/run.log | grep rep
f=$(ls $o/t_kernel_stats.csv $o/*/t_kernel_stats.csv 2>/dev/null | head -1)
python3 - "$f" <<'PY'
import csv, sys
for i, r in enumerate(csv.DictReader(open(sys.argv[1]))):
    import re
    m = re.search(r"(k_\w+)(<[^>]*>)?", r["Name"])
    n = (m.group(1) + (m.group(2) or "")) if m else r["Name"][:40]
    if i < 10: print("%-40s calls %6s total %9.2f ms avg %9.1f us  %5s%%" % (n, r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
# the last factorisation of the run: busy time per kernel and per queue against its span (overlap of the look-ahead stream)
python3 - <<'PY'
import csv, glob, collections, re
f = glob.glob("gpurun_out/psdtr/**/t_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda x: int(x["Start_Timestamp"]))
sc = [i for i, r in enumerate(rows) if "k_chol_scatter" in r["Kernel_Name"]]
sel = rows[sc[-1]:]
t0, t1 = int(sel[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in sel)
agg = collections.defaultdict(lambda: [0, 0.0])
byq = collections.defaultdict(float)
for r in sel:
    m = re.search(r"(k_\w+)", r["Kernel_Name"])
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg[m.group(1) if m else "?"][0] += 1; agg[m.group(1) if m else "?"][1] += d
    byq[r.get("Queue_Id", "?")] += d
print("span %.2f ms, sum of kernel durations %.2f ms" % ((t1 - t0) / 1e6, sum(a[1] for a in agg.values()) / 1e3))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]): print("  %-22s %5d launches %9.2f ms  avg %8.1f us" % (k, a[0], a[1] / 1e3, a[1] / a[0]))
print("  per queue:", {k: round(v / 1e3, 2) for k, v in byq.items()})
# union of busy intervals per queue overlapping
ev = []
for r in sel: ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort(); depth = 0; last = None; two = 0; one = 0
for t, dlt in ev:
    if last is not None:
        if depth >= 2: two += t - last
        elif depth == 1: one += t - last
    depth += dlt; last = t
ea = sorted(((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, int(r.get("Grid_Size", 0)) // 256) for r in sel if "extend_add" in r["Kernel_Name"])
if ea:
    import bisect
    tot = sum(d for d, _ in ea)
    for lim in (6, 12, 25, 50, 100, 1e9):
        part = [d for d, _ in ea if d < lim]
        print("  extend_add launches under %g us: %d, %.2f ms" % (lim, len(part), sum(part) / 1e3))
    print("  longest extend_add launches (us, workgroups):", [(round(d, 1), g) for d, g in ea[-8:]])
print("  time with >= 2 kernels in flight %.2f ms, exactly one %.2f ms, none %.2f ms" % (two / 1e6, one / 1e6, (t1 - t0 - two - one) / 1e6))
PY
rm -f $o/t_kernel_trace.csv $o/*/t_kernel_trace.csv
