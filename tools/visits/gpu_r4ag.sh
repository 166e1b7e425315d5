#!/bin/bash
# visit 4ag: the A/Bs of visits 4x and 4ae again, valid this time (csrc/build.py --variant used to rebuild the DEFAULT library from the edited source as
# well: both compared a build with itself), plus two more fp32 bounds: 3x3 convs without operand fetches after the first K tile, no epilogue at all
set -o pipefail
mkdir -p gpurun_out
L=yolo-v3-tf2_amd/lib
Y3_LIB_PATH=$PWD/$L/liby3hip_earlyres.so timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bf16" > gpurun_out/r4ag_tests_earlyres.txt 2>&1 || { tail -30 gpurun_out/r4ag_tests_earlyres.txt; exit 1; }
tail -1 gpurun_out/r4ag_tests_earlyres.txt
timeout -k 10 700 python tools/ab_libs.py $L/liby3hip.so $L/liby3hip_earlyres.so --dtype bf16 --batch 128 --rounds 3 > gpurun_out/r4ag_ab_bf16_earlyres.txt 2>&1 || { tail -20 gpurun_out/r4ag_ab_bf16_earlyres.txt; exit 1; }
grep -v amdgpu gpurun_out/r4ag_ab_bf16_earlyres.txt | tail -2
timeout -k 10 900 python tools/ab_libs.py $L/liby3hip.so $L/liby3hip_nofetch1x1.so $L/liby3hip_nofetch3.so $L/liby3hip_noepi.so --dtype f32 --batch 64 --rounds 3 > gpurun_out/r4ag_ab_f32_bounds.txt 2>&1 || { tail -20 gpurun_out/r4ag_ab_f32_bounds.txt; exit 1; }
grep -v amdgpu gpurun_out/r4ag_ab_f32_bounds.txt | tail -4
