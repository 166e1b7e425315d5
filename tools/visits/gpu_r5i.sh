#!/bin/bash
# visit 5i: fp32 prologue with the closed-form tap mask (mask), + prologue / epilogue at s_setprio 3 (maskprio) against the previous commit (base)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "conv_layers or every_tile_shape or network_grids or full_size_batch or chunk_major" > gpurun_out/5i_tests.log 2>&1 || { tail -40 gpurun_out/5i_tests.log; exit 1; }
tail -1 gpurun_out/5i_tests.log
L=yolo-v3-tf2_amd/lib
timeout -k 10 700 python tools/ab_libs.py $L/liby3hip_base.so $L/liby3hip_mask.so $L/liby3hip_maskprio.so --rounds 3 > gpurun_out/5i_ab.txt 2>&1 || { tail -20 gpurun_out/5i_ab.txt; exit 1; }
grep -v amdgpu gpurun_out/5i_ab.txt | tail -12
