#!/bin/bash
# which hardware queue every kernel of a session's set-up ran on, and when (kernel trace of tools/setup_timing.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/setupq
rm -rf $o; mkdir -p $o
rocprofv3 --kernel-trace --output-format csv -d $o -o t -- python3 tools/setup_timing.py > $o/run.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/setupq/**/t_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(rows[0].keys())
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last session creation: take the last 700 kernels whose name has chol / dense
sel = [r for r in rows if any(s in r["Kernel_Name"] for s in ("k_chol", "k_dense"))][-660:]
t0 = int(sel[0]["Start_Timestamp"])
byq = collections.defaultdict(list)
for r in sel:
    byq[(r.get("Queue_Id"), r.get("Stream_Id", ""))].append(((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r["Kernel_Name"].split("(")[0][-24:]))
for q, v in byq.items():
    print("queue/stream", q, "kernels", len(v), "first start %.0f us, last end %.0f us" % (v[0][0], v[-1][1]), "busy %.0f us" % sum(b - a for a, b, _ in v))
PY
rm -f $(find $o -name "t_kernel_trace.csv")
