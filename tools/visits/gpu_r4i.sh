#!/bin/bash
# visit 4i: weight-resident bf16 3x3 kernel (tile 32): tests, then bench bf16 with the heuristic... and the table A/B
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "weight_resident or bf16" > gpurun_out/r4i_tests.log 2>&1 || { tail -60 gpurun_out/r4i_tests.log; exit 1; }
tail -2 gpurun_out/r4i_tests.log
timeout -k 10 600 python tools/tune_tiles.py --dtype bf16 --batch 128 --tiles 30,31,5,10,8,22,32 --reps 3 > gpurun_out/r4i_sweep_bf16_resident.txt 2>&1 || { tail -30 gpurun_out/r4i_sweep_bf16_resident.txt; exit 1; }
grep -v amdgpu gpurun_out/r4i_sweep_bf16_resident.txt | awk 'NR==1 || /k3s1_c32|k3s1_c64|sum/'
