cd /root/repo
o=gpurun_out/sp_v2b.log; : > $o
timeout -k 10 300 python -m pytest tests/test_sparse_precond.py -m gpu -x -q >> $o 2>&1 || exit 1
timeout -k 10 120 python tools/bench_sp_lanes.py all >> $o 2>&1 || exit 1
for v in "400,160,40,12,0" "400,120,20,8,0" "500,200,20,8,0" "300,120,20,8,0" "400,160,30,8,0" "100000,160,20,8,0"; do
  DCORA_SP_LANES=$v timeout -k 10 120 python tools/bench_sp_lanes.py all >> $o 2>&1 || exit 1
done
