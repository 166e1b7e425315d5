#!/bin/bash
# visit 4y: the power wall in the running step: the same library, the same launches, on random data and on all-zero weights + images
# (nothing toggles in the matrix pipes or on the data paths) -- conv stack per step, bf16 (128 images) and fp32 (64 images)
set -o pipefail
mkdir -p gpurun_out
L=yolo-v3-tf2_amd/lib
for dt in bf16:128 f32:64; do
  d=${dt%%:*}; b=${dt##*:}
  for data in rand zeros rand zeros; do
    timeout -k 10 300 python tools/ab_libs.py $L/liby3hip.so --dtype $d --batch $b --rounds 1 --data $data > gpurun_out/r4y_tmp.txt 2>&1 || { tail -20 gpurun_out/r4y_tmp.txt; exit 1; }
    echo "$d b$b $data: $(grep -v amdgpu gpurun_out/r4y_tmp.txt | grep '^round')" | tee -a gpurun_out/r4y_power_wall.txt
  done
done
