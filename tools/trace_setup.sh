#!/bin/bash
# kernel trace of the creation of the headline session (5 agents of sphere2500, dense preconditioners built side by side
# on host threads): how much of the device work of the five builds overlaps
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/setuptr
rm -rf $o; mkdir -p $o
cat > $o/run.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests"))
import common, dcora_amd as da
from dcora_amd import capi
ds = common.product_dataset("sphere2500")
s = da.RbcdSession(ds, num_robots=5, r=5); s.close()
capi.lib().dcora_precond_cache_clear()
t0 = time.perf_counter(); s = da.RbcdSession(ds, num_robots=5, r=5); t1 = time.perf_counter()
print("session %.2f ms" % (1e3 * (t1 - t0)))
s.close()
PY
rocprofv3 --kernel-trace --output-format csv -d $o -o t -- python3 $o/run.py > $o/run.log 2>&1
grep session $o/run.log
python3 - <<'PY'
import csv, glob, collections, re
f = glob.glob("gpurun_out/setuptr/**/t_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda x: int(x["Start_Timestamp"]))
# the second creation: everything after the last but one k_dense_scatter group (5 per creation)
sc = [i for i, r in enumerate(rows) if "k_dense_scatter" in r["Kernel_Name"]]
sel = rows[sc[-5]:]
t0, t1 = int(sel[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in sel)
agg = collections.defaultdict(lambda: [0, 0.0]); byq = collections.defaultdict(float)
for r in sel:
    m = re.search(r"(k_\w+)", r["Kernel_Name"]); d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg[m.group(1) if m else "?"][0] += 1; agg[m.group(1) if m else "?"][1] += d; byq[r.get("Queue_Id", "?")] += d
print("span %.2f ms, %d launches, sum of kernel durations %.2f ms" % ((t1 - t0) / 1e6, len(sel), sum(a[1] for a in agg.values()) / 1e3))
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:10]: print("  %-26s %5d launches %8.2f ms  avg %7.1f us" % (k, a[0], a[1] / 1e3, a[1] / a[0]))
print("  per queue (ms):", {k: round(v / 1e3, 2) for k, v in byq.items()})
ev = []
for r in sel: ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort(); depth = 0; last = None; hist = collections.defaultdict(int)
for t, dlt in ev:
    if last is not None: hist[min(depth, 5)] += t - last
    depth += dlt; last = t
print("  time by kernels in flight (ms):", {k: round(v / 1e6, 2) for k, v in sorted(hist.items())})
PY
rm -f $o/t_kernel_trace.csv $o/*/t_kernel_trace.csv
