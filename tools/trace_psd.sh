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
rm -f $o/t_kernel_trace.csv $o/*/t_kernel_trace.csv
