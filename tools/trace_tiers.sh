#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/tierstr
rm -rf $o; mkdir -p $o
rocprofv3 --kernel-trace --stats --output-format csv -d $o -o t -- python3 tools/prof_tiers_level.py > $o/run.log 2>&1
tail -1 $o/run.log
python3 tools/kstats.py $o 14
rm -f $o/t_kernel_trace.csv $o/*/t_kernel_trace.csv
