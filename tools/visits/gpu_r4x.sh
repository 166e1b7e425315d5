#!/bin/bash
# visit 4x: bf16 generic kernel with the last K tile peeled and the first block's shortcut rows requested before its MFMAs:
# parity (bit-identical arithmetic), then alternating processes against the committed library
set -o pipefail
mkdir -p gpurun_out
L=yolo-v3-tf2_amd/lib
Y3_LIB_PATH=$PWD/$L/liby3hip_earlyres.so timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bf16" > gpurun_out/r4x_tests.txt 2>&1 || { tail -30 gpurun_out/r4x_tests.txt; exit 1; }
tail -1 gpurun_out/r4x_tests.txt
timeout -k 10 700 python tools/ab_libs.py $L/liby3hip.so $L/liby3hip_earlyres.so --dtype bf16 --batch 128 --rounds 4 > gpurun_out/r4x_ab_bf16_earlyres.txt 2>&1 || { tail -20 gpurun_out/r4x_ab_bf16_earlyres.txt; exit 1; }
grep -v amdgpu gpurun_out/r4x_ab_bf16_earlyres.txt | tail -3
for l in liby3hip liby3hip_earlyres; do
  Y3_LIB_PATH=$PWD/$L/$l.so timeout -k 10 300 python tools/tune_tiles.py --dtype bf16 --batch 64 --tiles 24 --reps 3 > gpurun_out/r4x_sweep_$l.txt 2>&1 || { tail -20 gpurun_out/r4x_sweep_$l.txt; exit 1; }
  echo "== $l"; grep -v amdgpu gpurun_out/r4x_sweep_$l.txt | grep -E "k3s1_c(128|256|512)|sum" | awk '{c[$2]++; if (c[$2] <= 1) print}'
done
