#!/bin/bash
# launch sequence of one tCG iteration of the first CORA level of tiers.pyfg
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/tierstl
rm -rf $o; mkdir -p $o
rocprofv3 --kernel-trace --output-format csv -d $o -o t -- python3 tools/prof_tiers_level.py > $o/run.log 2>&1
grep "tCG" $o/run.log
python3 tools/tcg_timeline.py $o ${1:-k_tcg_update1}
rm -f $o/t_kernel_trace.csv $o/*/t_kernel_trace.csv
