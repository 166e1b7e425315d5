#!/bin/bash
# visit 4w: timing-only bound: bf16 generic kernel without the shortcut operand (no loads, no adds; wrong results) -- what hiding the
# shortcut loads behind the K loop could return at most
set -o pipefail
mkdir -p gpurun_out
L=yolo-v3-tf2_amd/lib
timeout -k 10 700 python tools/ab_libs.py $L/liby3hip.so $L/liby3hip_nores.so --dtype bf16 --batch 128 --rounds 3 > gpurun_out/r4w_ab_bf16_nores.txt 2>&1 || { tail -20 gpurun_out/r4w_ab_bf16_nores.txt; exit 1; }
grep -v amdgpu gpurun_out/r4w_ab_bf16_nores.txt | tail -3
for l in liby3hip liby3hip_nores; do
  Y3_LIB_PATH=$PWD/$L/$l.so timeout -k 10 300 python tools/tune_tiles.py --dtype bf16 --batch 64 --tiles 24 --reps 3 > gpurun_out/r4w_sweep_$l.txt 2>&1 || { tail -20 gpurun_out/r4w_sweep_$l.txt; exit 1; }
  echo "== $l"; grep -v amdgpu gpurun_out/r4w_sweep_$l.txt | grep -E "k3s1_c(128|256|512)|sum" | awk '{c[$2]++; if (c[$2] <= 1) print}'
done
