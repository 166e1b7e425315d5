#!/bin/bash
# visit 4x: steady-state coordinate descent over the fp32 table of 64 x 608^2 with two lanes; then the bf16 table once more on HEAD
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python tools/tune_steady.py --dtype f32 --batch 64 --image-size 608 --lanes 2 --steps 8 --write f32_b64_s608.json > gpurun_out/4x_tune_steady_f32_s608.txt 2>&1 || { tail -20 gpurun_out/4x_tune_steady_f32_s608.txt; exit 1; }
grep -v "keeps tile" gpurun_out/4x_tune_steady_f32_s608.txt | grep -v amdgpu
cp yolo-v3-tf2_amd/tuning/f32_b64_s608.json gpurun_out/4x_f32_b64_s608.json
