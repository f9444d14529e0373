#!/bin/bash
# per-launch durations of one application of the sparse replay from a rocprofv3 kernel trace: trace_sp.sh [lattice|sphere|tiers]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/sptr
rm -rf $o; mkdir -p $o
rocprofv3 --kernel-trace --output-format csv -d $o -o t -- python3 tools/bench_sp.py ${1:-lattice} > $o/run.log 2>&1
cat $o/run.log | tail -2
python3 tools/sp_trace.py $o
rm -rf $o/*/t_kernel_trace.csv $o/t_kernel_trace.csv
