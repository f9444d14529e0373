#!/bin/bash
# rocprofv3 kernel trace of exactly the timed loop of the headline workload; prints the top kernels
export TMPDIR=/tmp
out=gpurun_out/prof_${1:-x}
rm -rf $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 bench.py --steps 300 --warmup 30 --headline-only > $out.log 2>&1
python3 tools/kstats.py $out ${2:-4}
