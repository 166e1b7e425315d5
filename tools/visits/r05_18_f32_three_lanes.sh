#!/bin/bash
# round 5, visit 18: the fp32 headline (64 x 416^2) on THREE lanes: lanes sweep with the shipped table, then the steady-state tuner in the three-lane regime
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python tools/lanes_sweep.py --dtype f32 --batch 64 --lanes 2,3,2,3 2>/dev/null | grep -v amdgpu | tee gpurun_out/r05_18_lanes_f32.txt
timeout -k 10 1000 python tools/tune_steady.py --dtype f32 --batch 64 --lanes 3 --write f32_b64_s416_lanes3.json > gpurun_out/r05_18_tune_steady_f32_lanes3.txt 2>&1 || { tail -20 gpurun_out/r05_18_tune_steady_f32_lanes3.txt; exit 1; }
grep -v amdgpu gpurun_out/r05_18_tune_steady_f32_lanes3.txt | grep -e "->" -e start -e final
cp yolo-v3-tf2_amd/tuning/f32_b64_s416_lanes3.json gpurun_out/r05_18_f32_b64_s416_lanes3.json
