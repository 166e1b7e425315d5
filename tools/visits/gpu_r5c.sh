#!/bin/bash
# visit 5c: compiler scheduling strategies for the two conv kernels (-mllvm -amdgpu-sched-strategy=max-ilp / iterative-ilp,
# -amdgpu-schedule-metric-bias=0) against the default build, same box, alternating processes (tools/ab_libs.py)
set -o pipefail
mkdir -p gpurun_out
L=yolo-v3-tf2_amd/lib
timeout -k 10 500 python tools/ab_libs.py $L/liby3hip.so $L/liby3hip_ilp.so $L/liby3hip_itilp.so $L/liby3hip_bias.so --rounds 2 > gpurun_out/5c_ab_f32.txt 2>&1 || { tail -20 gpurun_out/5c_ab_f32.txt; exit 1; }
tail -4 gpurun_out/5c_ab_f32.txt
timeout -k 10 500 python tools/ab_libs.py $L/liby3hip.so $L/liby3hip_ilp.so $L/liby3hip_itilp.so $L/liby3hip_bias.so --dtype bf16 --batch 128 --rounds 2 > gpurun_out/5c_ab_bf16.txt 2>&1 || { tail -20 gpurun_out/5c_ab_bf16.txt; exit 1; }
tail -4 gpurun_out/5c_ab_bf16.txt
