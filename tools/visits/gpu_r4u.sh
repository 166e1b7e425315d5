#!/bin/bash
# visit 4u: the bf16 profile again (tools/profile.sh now runs without the parity gate's extra forward), the bf16 lane sweep on the round-4 library
set -o pipefail
mkdir -p gpurun_out
bash tools/profile.sh r04_bf16 --dtype bf16 --batch 128 > gpurun_out/r4u_profile.log 2>&1 || { tail -20 gpurun_out/r4u_profile.log; exit 1; }
tail -3 gpurun_out/r4u_profile.log
timeout -k 10 400 python tools/lanes_sweep.py --dtype bf16 --batch 128 --lanes 1,2,3,4 > gpurun_out/r4u_lanes_bf16.txt 2>&1 || { tail -20 gpurun_out/r4u_lanes_bf16.txt; exit 1; }
grep -v amdgpu gpurun_out/r4u_lanes_bf16.txt | tail -8
