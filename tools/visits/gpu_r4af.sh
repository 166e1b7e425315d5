#!/bin/bash
# visit 4af: timing-only bounds on the fp32 two-lane step: (a) the 3x3 convs without operand fetches after their first K tile, (b) the generic kernel
# without its epilogue (no shortcut loads, no arithmetic, no stores) -- wrong results both
set -o pipefail
mkdir -p gpurun_out
L=yolo-v3-tf2_amd/lib
timeout -k 10 800 python tools/ab_libs.py $L/liby3hip.so $L/liby3hip_nofetch3.so $L/liby3hip_noepi.so --dtype f32 --batch 64 --rounds 3 > gpurun_out/r4af_ab_f32_bounds.txt 2>&1 || { tail -20 gpurun_out/r4af_ab_f32_bounds.txt; exit 1; }
grep -v amdgpu gpurun_out/r4af_ab_f32_bounds.txt | tail -3
