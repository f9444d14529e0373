#!/bin/bash
# launch sequence of the RBCD iterations of the headline loop (most frequent period between two k_eval_finish launches)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
o=gpurun_out/hltl
rm -rf $o; mkdir -p $o
rocprofv3 --kernel-trace --output-format csv -d $o -o t -- python3 bench.py --steps 300 --warmup 30 --headline-only > $o/run.log 2>&1
python3 tools/tcg_timeline.py $o ${1:-k_eval_finish}
rm -f $o/t_kernel_trace.csv $o/*/t_kernel_trace.csv
