#!/bin/bash
# visit 4ae: timing-only bound: fp32 generic kernel, 1x1 convs only, operand fetches of K tiles 2.. removed (wrong results): what the memory side of the
# 1x1 layers costs the running two-lane step at most
set -o pipefail
mkdir -p gpurun_out
L=yolo-v3-tf2_amd/lib
timeout -k 10 700 python tools/ab_libs.py $L/liby3hip.so $L/liby3hip_nofetch1x1.so --dtype f32 --batch 64 --rounds 3 > gpurun_out/r4ae_ab_f32_nofetch1x1.txt 2>&1 || { tail -20 gpurun_out/r4ae_ab_f32_nofetch1x1.txt; exit 1; }
grep -v amdgpu gpurun_out/r4ae_ab_f32_nofetch1x1.txt | tail -2
