#!/bin/bash
# round 5, visit 17: bf16 128 x 416^2: the shipped two-lane table against (a) the same with the 1x1 512 -> 256 @26 convs on the 256x256 16-wave tile (24) and (b) the
# three-lane table the steady-state tuner produced (visit 16: that one signature changed), alternating child processes
set -o pipefail
mkdir -p gpurun_out
L=yolo-v3-tf2_amd/lib/liby3hip.so
timeout -k 10 900 python tools/ab_libs.py "$L" "$L@tools/tables/bf16_b128_s416_lanes2_1x1at26_tile24.json" "$L@tools/tables/bf16_b128_s416_lanes3.json" --dtype bf16 --batch 128 --rounds 4 > gpurun_out/r05_17_ab_bf16_tables.txt 2>&1 || { tail -20 gpurun_out/r05_17_ab_bf16_tables.txt; exit 1; }
grep -v amdgpu gpurun_out/r05_17_ab_bf16_tables.txt | tail -16
