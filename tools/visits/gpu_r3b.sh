#!/bin/bash
# round-3 GPU visit B: A/B of the lean pointwise prologue + straight-line epilogue (conv_f32.hip) against the r02 forms
set -o pipefail
mkdir -p gpurun_out
L=yolo-v3-tf2_amd/lib
timeout -k 10 600 python tools/ab_libs.py $L/liby3hip_base.so $L/liby3hip.so --rounds 3 > gpurun_out/ab_e1.log 2>&1 || { tail -20 gpurun_out/ab_e1.log; exit 1; }
tail -8 gpurun_out/ab_e1.log
Y3_LIB_PATH=$PWD/$L/liby3hip_base.so timeout -k 10 300 python tools/tune_tiles.py --tiles 10,11,17,26,27,31,32 --reps 3 > gpurun_out/sweep_base.log 2>&1 || { tail gpurun_out/sweep_base.log; exit 1; }
timeout -k 10 300 python tools/tune_tiles.py --tiles 10,11,17,26,27,31,32 --reps 3 > gpurun_out/sweep_e1.log 2>&1 || { tail gpurun_out/sweep_e1.log; exit 1; }
tail -3 gpurun_out/sweep_base.log gpurun_out/sweep_e1.log
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "conv or network or full_size" > gpurun_out/pytest_e1.log 2>&1; tail -3 gpurun_out/pytest_e1.log
