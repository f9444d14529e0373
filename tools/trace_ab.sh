#!/bin/bash
# scratch: kernel traces of the headline loop under both tCG pacing modes (run on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/ab
mkdir -p $o
rocprofv3 --kernel-trace --stats --output-format csv -d $o/paced -o t -- python3 bench.py --headline-only > $o/paced.log 2>&1 &&
DCORA_PC_PACING=lookahead rocprofv3 --kernel-trace --stats --output-format csv -d $o/look -o t -- python3 bench.py --headline-only > $o/look.log 2>&1
for d in paced look; do
  f=$(ls $o/$d/t_kernel_stats.csv $o/$d/*/t_kernel_stats.csv 2>/dev/null | head -1)
  echo "== $d $f"
  [ -n "$f" ] && head -8 "$f"
done
# keep the traces small enough to travel back: drop everything but the two traces
find $o -name "*.csv" ! -name "t_kernel_trace.csv" ! -name "t_kernel_stats.csv" -delete
du -sh $o
python3 - <<'PY'
import csv, collections, glob
for d in ("paced", "look"):
    f = glob.glob("gpurun_out/ab/%s/**/t_kernel_trace.csv" % d, recursive=True)[0]
    h = collections.defaultdict(lambda: [0, 0, 0.0, 0.0])
    rows = list(csv.DictReader(open(f)))
    for x in rows:
        n = x["Kernel_Name"]
        key = "pc" if "k_fused_pc" in n else "hess" if "k_fused_hess" in n else None
        if not key: continue
        dur = int(x["End_Timestamp"]) - int(x["Start_Timestamp"])
        e = h[key]
        if dur < (5400 if key == "pc" else 3800): e[0] += 1; e[2] += dur
        else: e[1] += 1; e[3] += dur
    # idle time between consecutive kernels over the whole trace
    ts = sorted((int(x["Start_Timestamp"]), int(x["End_Timestamp"])) for x in rows)
    gaps = [b[0] - a[1] for a, b in zip(ts, ts[1:]) if 0 < b[0] - a[1] < 200000]
    print(d, {k: (v[0], round(v[2] / max(v[0], 1)), v[1], round(v[3] / max(v[1], 1))) for k, v in h.items()},
          "gaps: n=%d mean=%.0f ns total=%.1f ms; busy=%.1f ms" % (len(gaps), sum(gaps) / len(gaps), sum(gaps) / 1e6, sum(e - s for s, e in ts) / 1e6))
PY
rm -rf gpurun_out/ab/*/t_kernel_trace.csv gpurun_out/ab/*/*/t_kernel_trace.csv
