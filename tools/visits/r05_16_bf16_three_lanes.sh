#!/bin/bash
# round 5, visit 16: bf16 128 x 416^2 on THREE lanes (r04_lanes_sweep_bf16.txt: 9.163 vs 9.201 ms with the two-lane table): steady-state re-tune in that regime, then 2 vs 3 lanes
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python tools/lanes_sweep.py --dtype bf16 --batch 128 --lanes 2,3,2,3 2>/dev/null | grep -v amdgpu | tee gpurun_out/r05_16_lanes_bf16.txt
timeout -k 10 900 python tools/tune_steady.py --dtype bf16 --batch 128 --lanes 3 --write bf16_b128_s416_lanes3.json > gpurun_out/r05_16_tune_steady_bf16_lanes3.txt 2>&1 || { tail -20 gpurun_out/r05_16_tune_steady_bf16_lanes3.txt; exit 1; }
grep -v amdgpu gpurun_out/r05_16_tune_steady_bf16_lanes3.txt | grep -e "->" -e start -e final
cp yolo-v3-tf2_amd/tuning/bf16_b128_s416_lanes3.json gpurun_out/r05_16_bf16_b128_s416_lanes3.json
