#!/bin/bash
# visit 5h: prologue / epilogue of the fp32 conv kernel at s_setprio 3 (K loop at 0): A/B against the shipped build (tools/ab_libs.py), then
# the phase stamps of one layer with the priorities in
set -o pipefail
mkdir -p gpurun_out
L=yolo-v3-tf2_amd/lib
timeout -k 10 500 python tools/ab_libs.py $L/liby3hip.so $L/liby3hip_prio.so --rounds 3 > gpurun_out/5h_ab_prio.txt 2>&1 || { tail -20 gpurun_out/5h_ab_prio.txt; exit 1; }
grep -v amdgpu gpurun_out/5h_ab_prio.txt | tail -8
export Y3_LIB_PATH=$PWD/$L/liby3hip_prio_stamps.so
timeout -k 10 200 python tools/phase_stamps.py --dtype f32 --batch 64 --cin 128 --cout 256 --s 52 --tile 31 2>&1 | grep -v amdgpu.ids > gpurun_out/5h_phase_stamps_prio.txt || { tail gpurun_out/5h_phase_stamps_prio.txt; exit 1; }
cat gpurun_out/5h_phase_stamps_prio.txt
