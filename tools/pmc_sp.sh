#!/bin/bash
# SQ counters of the sparse replay's kernels (lattice agent): where the wave cycles go
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/sppmc
rm -rf $o; mkdir -p $o
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $o -o t -- python3 tools/bench_sp.py lattice > $o/run.log 2>&1
tail -2 $o/run.log
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/sppmc/**/t_counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0][-40:]
    agg[n][r["Counter_Name"]] += float(r["Counter_Value"]); 
    if r["Counter_Name"] == "SQ_WAVES": cnt[n] += 1
for n, c in agg.items():
    if cnt[n] < 50: continue
    k = cnt[n]
    print(n, "dispatches", k, {a: round(b / k, 1) for a, b in c.items()})
PY
rm -rf $o/*/t_counter_collection.csv $o/*/t_kernel_trace.csv
