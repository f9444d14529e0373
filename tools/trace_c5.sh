#!/bin/bash
# scratch: kernel trace of the C5 loop, summarised on the box (the trace itself is too large to travel back)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/c5tr
rm -rf $o; mkdir -p $o
rocprofv3 --kernel-trace --stats --output-format csv -d $o -o t -- python3 tools/bench_c5.py 60 > $o/run.log 2>&1
tail -1 $o/run.log | cut -c1-400
python3 - <<'PY'
import csv, collections, glob, re
f = glob.glob("gpurun_out/c5tr/**/t_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda x: int(x["Start_Timestamp"]))
# the timed loop = the last 60 k_eval_finish launches
ev = [i for i, x in enumerate(rows) if "k_eval_finish" in x["Kernel_Name"]]
lo = ev[-61] if len(ev) > 61 else 0
sel = rows[lo:ev[-1] + 1]
span = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e3
agg = collections.defaultdict(lambda: [0, 0.0, 0, 0.0])
busy = 0.0
for x in sel:
    n = re.sub(r"<.*", "", x["Kernel_Name"].replace("void ", "").replace("dcora::", "").replace("(anonymous namespace)::", ""))
    n = n.split("(")[0]
    d = (int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e3
    a = agg[n]
    busy += d
    if d < 5.2: a[2] += 1; a[3] += d
    else: a[0] += 1; a[1] += d
print("window: %.1f us per RBCD iteration over 60 iterations; busy %.1f us, gaps %.1f us" % (span / 60, busy / 60, (span - busy) / 60))
print("%-28s %8s %9s %9s | %8s %9s" % ("kernel", "n/iter", "avg us", "us/iter", "short/it", "us/iter"))
for n, a in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][3])):
    print("%-28s %8.2f %9.2f %9.1f | %8.2f %9.1f" % (n[:28], a[0] / 60, a[1] / max(a[0], 1), a[1] / 60, a[2] / 60, a[3] / 60))
PY
rm -rf $o/*/t_kernel_trace.csv $o/t_kernel_trace.csv
