cd /root/repo
o=gpurun_out/sp_v2e.log; : > $o
timeout -k 10 120 python tools/bench_sp_lanes.py all >> $o 2>&1 || exit 1
for v in 96 192 384; do
  echo "leaf $v" >> $o
  DCORA_ND_LEAF=$v timeout -k 10 120 python tools/bench_sp_lanes.py all >> $o 2>&1 || exit 1
done
echo "top 4096" >> $o
DCORA_ND_TOP=4096 timeout -k 10 120 python tools/bench_sp_lanes.py all >> $o 2>&1 || exit 1
echo "top 2048" >> $o
DCORA_ND_TOP=2048 timeout -k 10 120 python tools/bench_sp_lanes.py all >> $o 2>&1 || exit 1
