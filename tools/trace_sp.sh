#!/bin/bash
# per-launch durations of one application of the sparse replay (lattice agent) from a rocprofv3 kernel trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/sptr${1:-}
rm -rf $o; mkdir -p $o
rocprofv3 --kernel-trace --output-format csv -d $o -o t -- python3 tools/bench_sp.py lattice > $o/run.log 2>&1
cat $o/run.log | tail -3
python3 tools/sp_trace.py $o
rm -rf $o/*/t_kernel_trace.csv $o/t_kernel_trace.csv
