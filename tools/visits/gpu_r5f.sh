#!/bin/bash
# visit 5f: steady-state coordinate descent over the fp32 table of 32 x 416^2 (BASELINE config 2 geometry), one lane
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python tools/tune_steady.py --dtype f32 --batch 32 --steps 25 --write f32_b32_s416.json > gpurun_out/5f_tune_steady_f32_b32.txt 2>&1 || { tail -20 gpurun_out/5f_tune_steady_f32_b32.txt; exit 1; }
grep -v "keeps tile" gpurun_out/5f_tune_steady_f32_b32.txt | grep -v amdgpu
cp yolo-v3-tf2_amd/tuning/f32_b32_s416.json gpurun_out/5f_f32_b32_s416.json
timeout -k 10 300 python tools/bench_configs.py > gpurun_out/5f_config2.txt 2>&1; tail -4 gpurun_out/5f_config2.txt
