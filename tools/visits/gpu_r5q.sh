#!/bin/bash
# visit 5q: border epilogue four elements at a time + tile 31 with the register budget of 6 waves per SIMD (72 VGPRs, no scratch: six workgroups per CU) against the previous commit
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "conv_layers or every_tile_shape or network_grids or full_size_batch or chunk_major or lanes_bit" > gpurun_out/5q_tests.log 2>&1 || { tail -40 gpurun_out/5q_tests.log; exit 1; }
tail -1 gpurun_out/5q_tests.log
L=yolo-v3-tf2_amd/lib
timeout -k 10 700 python tools/ab_libs.py $L/liby3hip_base.so $L/liby3hip.so --rounds 4 > gpurun_out/5q_ab.txt 2>&1 || { tail -20 gpurun_out/5q_ab.txt; exit 1; }
grep -v amdgpu gpurun_out/5q_ab.txt | tail -10
