#!/bin/bash
# visit u: chunk-major K order of the fp32 3x3 convs: time per launch and traffic beyond L2, per chunk size
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/u_time.log
for shape in "512 1024 13" "256 512 26" "128 256 52" "64 128 104"; do
  set -- $shape
  for ck in 0 32 64 128; do
    Y3_K_CHUNK=$ck timeout -k 10 120 python tools/one_conv.py --dtype f32 --cin $1 --cout $2 --s $3 --batch 64 --reps 30 2>/dev/null | sed "s/^/ck=$ck  /" >> gpurun_out/u_time.log || { echo "one_conv failed ($shape ck=$ck)"; exit 1; }
  done
done
cat gpurun_out/u_time.log
for ck in 0 32 64; do
  Y3_K_CHUNK=$ck LINES_OUT=30 bash tools/pmc_one_conv.sh u_ck$ck --dtype f32 --cin 512 --cout 1024 --s 13 --batch 64 > gpurun_out/u_pmc_ck$ck.log 2>&1 || { tail -20 gpurun_out/u_pmc_ck$ck.log; exit 1; }
  grep -i "fetch\|write\|TCC" gpurun_out/u_pmc_ck$ck.log | head -8
done
