#!/bin/bash
# Counters of the block Q-apply (k_spmm_bsrq) on the whole 100k lattice, warm launches (one set re-read) and cold ones
# (four sets in turn): fabric bytes, L2 hits / misses, where the wave cycles go.  One rocprofv3 --pmc pass per group
# (FETCH_SIZE and WRITE_SIZE do not fit one pass); writes gpurun_out/qapply_pmc/summary.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/qapply_pmc
rm -rf $o; mkdir -p $o
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $o/p$i -o t -- python3 tools/bench_qapply.py 5 > $o/p$i.log 2>&1
  tail -1 $o/p$i.log
done
python3 - <<'PY' > gpurun_out/qapply_pmc/summary.txt
import csv, glob, collections, statistics
print("# k_spmm_bsrq on the 100k-pose lattice (r = 5): rocprofv3 --pmc, per dispatch; warm = the 100 back-to-back launches on one")
print("# (Q, X, Y) set, cold = the 96 launches that rotate over four sets (497 MB between two uses of a set)")
for f in sorted(glob.glob("gpurun_out/qapply_pmc/p*/**/t_counter_collection.csv", recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    order = []
    for r in csv.DictReader(open(f)):
        if "k_spmm_bsrq" not in r["Kernel_Name"]:
            continue
        d = int(r["Dispatch_Id"])
        if d not in per:
            order.append(d)
        per[d][r["Counter_Name"]] += float(r["Counter_Value"])
    order.sort()
    if len(order) < 150:
        print("# %s: only %d dispatches" % (f, len(order)))
        continue
    cold, warm = order[-96:], order[-196:-96]
    for name in sorted(per[order[-1]].keys()):
        w = statistics.median(per[d][name] for d in warm)
        c = statistics.median(per[d][name] for d in cold)
        print("%-20s warm %16.1f   cold %16.1f" % (name, w, c))
PY
cat gpurun_out/qapply_pmc/summary.txt
find $o -name "*counter_collection.csv" -delete; find $o -name "*kernel_trace.csv" -delete
