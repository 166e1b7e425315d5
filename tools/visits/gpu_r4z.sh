#!/bin/bash
# visit 4z: steady-state coordinate descent over the bf16 table (128 images, two lanes) on the round-4 library, the tap-row-reuse tiles
# (33..35) among the candidates
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python tools/tune_steady.py --dtype bf16 --batch 128 --steps 20 > gpurun_out/r4z_tune_steady_bf16_b128.txt 2>&1 || { tail -20 gpurun_out/r4z_tune_steady_bf16_b128.txt; exit 1; }
grep -v amdgpu gpurun_out/r4z_tune_steady_bf16_b128.txt | grep -E "start|final|->" 
