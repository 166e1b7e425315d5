#!/bin/bash
# visit 4f: fp32 lanes sweep and a steady-state re-tune of the 64 x 416^2 table on the round-4 library
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python tools/lanes_sweep.py --dtype f32 --lanes 1,2,3,4 > gpurun_out/r4f_lanes_sweep_f32.txt 2>&1 || { tail -20 gpurun_out/r4f_lanes_sweep_f32.txt; exit 1; }
grep -v amdgpu gpurun_out/r4f_lanes_sweep_f32.txt | tail -8
cp yolo-v3-tf2_amd/tuning/f32_b64_s416.json gpurun_out/r4f_f32_b64_s416_before.json
timeout -k 10 1000 python tools/tune_steady.py --dtype f32 --batch 64 --steps 15 --write f32_b64_s416.json > gpurun_out/r4f_tune_steady_f32.txt 2>&1 || { tail -20 gpurun_out/r4f_tune_steady_f32.txt; exit 1; }
grep -v "keeps tile" gpurun_out/r4f_tune_steady_f32.txt | grep -v amdgpu
cp yolo-v3-tf2_amd/tuning/f32_b64_s416.json gpurun_out/r4f_f32_b64_s416.json
