#!/bin/bash
# round 5, visit 5: the step with the fused 1x1 tails on / off (Y3_TAIL_FUSION; off = the launch structure of the round's base build), bf16, 128 x 416^2, alternating
set -o pipefail
mkdir -p gpurun_out
L=yolo-v3-tf2_amd/lib
timeout -k 10 700 python tools/ab_libs.py "$L/liby3hip.so%Y3_TAIL_FUSION=0" "$L/liby3hip.so%Y3_TAIL_FUSION=1" --dtype bf16 --batch 128 --rounds 4 > gpurun_out/r05_05_ab_bf16_tail.txt 2>&1 || { tail -20 gpurun_out/r05_05_ab_bf16_tail.txt; exit 1; }
grep -v amdgpu gpurun_out/r05_05_ab_bf16_tail.txt | tail -12
