#!/bin/bash
# visit 4s: tap-row reuse in the running pipeline (two lanes): the shipped table against the same table with tile 33 on the six
# 3x3 / stride-1 signatures with Cin % 128 == 0, and the timing-only build without the halo pass
set -o pipefail
mkdir -p gpurun_out
L=yolo-v3-tf2_amd/lib
T=tools/tables/bf16_b128_s416_rs.json
timeout -k 10 800 python tools/ab_libs.py $L/liby3hip.so $L/liby3hip.so@$T $L/liby3hip_rs_NO_EXTRA.so@$T --dtype bf16 --batch 128 --rounds 3 > gpurun_out/r4s_ab_bf16_rs.txt 2>&1 || { tail -20 gpurun_out/r4s_ab_bf16_rs.txt; exit 1; }
grep -v amdgpu gpurun_out/r4s_ab_bf16_rs.txt | tail -12
timeout -k 10 400 python tools/tune_tiles.py --dtype bf16 --batch 64 --tiles 24,27,29,33,34,35 --reps 3 > gpurun_out/r4s_sweep_b64.txt 2>&1 || { tail -20 gpurun_out/r4s_sweep_b64.txt; exit 1; }
grep -v amdgpu gpurun_out/r4s_sweep_b64.txt | grep -E "k3s1_c(128|256|512)|conv  shape|sum" | awk '{c[$2]++; if (c[$2] <= 1 || $1 == "conv") print}'
