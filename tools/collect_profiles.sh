#!/bin/bash
# collects the round's rocprofv3 summaries into gpurun_out/<tag> (run on the GPU box; tools/assemble_profiles.py <tag>
# copies what is quoted into profiles/): collect_profiles.sh [tag] [part ...]
# parts (a gpurun call lasts at most 20 minutes: one or two parts per call; gpurun_out/<tag> is merged back per call):
#   traces  kernel summaries of the headline loop and of the whole default run
#   pmc     FETCH_SIZE / WRITE_SIZE passes over tools/pmc_run.py
#   bench   the default bench.py run (compact line + full detail)
#   ranks   bench.py --gpus 2 rehearsed on ONE GPU over gloo
export TMPDIR=/tmp
tag=${1:-r05}
shift
parts=${@:-traces pmc bench ranks}
o=gpurun_out/$tag
mkdir -p $o
for part in $parts; do
  case $part in
    traces)
      rm -rf $o/headline $o/default
      rocprofv3 --kernel-trace --stats --output-format csv -d $o/headline -o t -- python3 bench.py --headline-only > $o/headline.log 2>&1
      python3 tools/kstats.py $o/headline 20 > $o/headline_top.txt
      rocprofv3 --kernel-trace --stats --output-format csv -d $o/default -o t -- python3 bench.py --no-cpu-baseline --no-coloured > $o/default.log 2>&1
      python3 tools/kstats.py $o/default 30 > $o/default_top.txt
      ;;
    pmc)
      rm -rf $o/pmc_fetch $o/pmc_write
      rocprofv3 --pmc FETCH_SIZE --output-format csv -d $o/pmc_fetch -o t -- python3 tools/pmc_run.py > $o/pmc_fetch.log 2>&1
      rocprofv3 --pmc WRITE_SIZE --output-format csv -d $o/pmc_write -o t -- python3 tools/pmc_run.py > $o/pmc_write.log 2>&1
      for c in FETCH_SIZE WRITE_SIZE; do
        d=$([ $c = FETCH_SIZE ] && echo pmc_fetch || echo pmc_write)
        f=$(find $o/$d -name "*counter_collection.csv" | head -1)
        python3 tools/pmc_summarize.py $f $c > $o/pmc_${c}_summary.csv
      done
      ;;
    bench)
      python3 bench.py > $o/bench_line.json 2> $o/bench.err
      cp gpurun_out/bench_detail.json $o/bench.json
      echo "bench done" >&2
      ;;
    ranks)
      # the N > 1 path rehearsed on ONE GPU (2 ranks over gloo, both on device 0): not a scaling figure
      DCORA_DIST_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 > $o/bench_2ranks_line.json 2> $o/bench_2ranks.err
      cp gpurun_out/bench_detail.json $o/bench_2ranks_on_one_gpu.json
      ;;
  esac
  echo "part $part done" >&2
done
# keep the merged output small
find $o -name "*kernel_trace.csv" -delete; find $o -name "*counter_collection.csv" -delete
ls -la $o
