#!/bin/bash
# visit 4d: table-driven prologue + compile-time kernel size in the classic fp32 conv kernel against the arithmetic prologue (-DY3_AB_ARITH_PROLOGUE)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "conv_layers or every_tile_shape or network_grids or full_size_batch_properties or chunk_major or lanes_bit or upsample_concat or xcd_blocked or persistent or end_to_end or 608_grids" > gpurun_out/r4d_tests.log 2>&1 || { tail -40 gpurun_out/r4d_tests.log; exit 1; }
tail -1 gpurun_out/r4d_tests.log
L=yolo-v3-tf2_amd/lib
timeout -k 10 700 python tools/ab_libs.py $L/liby3hip_arith.so $L/liby3hip.so --rounds 4 > gpurun_out/r4d_ab_rowtab.txt 2>&1 || { tail -20 gpurun_out/r4d_ab_rowtab.txt; exit 1; }
grep -v amdgpu gpurun_out/r4d_ab_rowtab.txt | tail -12
