#!/bin/bash
# scratch: per-launch efficiency of k_chol_syrk in the factorisation of the whole 100k lattice
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/choltr
rm -rf $o; mkdir -p $o
DCORA_CHOL_PLAN_DUMP=$o/plan.txt rocprofv3 --kernel-trace --output-format csv -d $o -o t -- python3 tools/prof_psd.py lattice > $o/run.log 2>&1
grep rep $o/run.log | tail -1
python3 - <<'PY'
import csv, glob, collections
plan = [l.split() for l in open("gpurun_out/choltr/plan.txt")]
syrk_plan = [(int(p[2]), int(p[3]), int(p[4]), int(p[5]), int(p[7]), float(p[8])) for p in plan if p[0] == "3"]
f = glob.glob("gpurun_out/choltr/**/t_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda x: int(x["Start_Timestamp"]))
syrk = [r for r in rows if "k_chol_syrk" in r["Kernel_Name"]]
n = len(syrk_plan)
last = syrk[-n:]   # the last factorisation of the run
assert len(last) == n
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
worst = []
for (gx, gy, j0, kcap, ncb, flop), r in zip(syrk_plan, last):
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    kind = "in-panel" if ncb > 0 else "trailing"
    size = "wg<256" if gx * gy < 256 else "wg<2048" if gx * gy < 2048 else "wg>=2048"
    a = agg[(kind, size)]
    a[0] += 1; a[1] += us; a[2] += flop
    worst.append((us, kind, gx, gy, kcap - j0, flop / us / 1e6 if us else 0))
tot_us = sum(a[1] for a in agg.values()); tot_f = sum(a[2] for a in agg.values())
print("syrk: %d launches, %.1f ms, %.1f Gflop, %.1f Tflop/s" % (n, tot_us / 1e3, tot_f / 1e9, tot_f / tot_us / 1e6))
for k, a in sorted(agg.items()):
    print("%-10s %-9s launches %4d  %7.2f ms  %8.1f Gflop  %6.1f Tflop/s" % (k[0], k[1], a[0], a[1] / 1e3, a[2] / 1e9, a[2] / max(a[1], 1e-9) / 1e6))
print("longest launches (us, kind, gx, gy, K, Tflop/s):")
for w in sorted(worst, reverse=True)[:12]: print("  %8.1f %-9s %6d %4d %4d %6.1f" % w)
PY
rm -f $o/t_kernel_trace.csv $o/*/t_kernel_trace.csv
