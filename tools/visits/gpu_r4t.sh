#!/bin/bash
# visit 4t: tap-row reuse with one address base per thread (fewer registers in the 128-register 16-wave tile): parity, then in the
# running pipeline against the shipped table; timing-only builds without the halo pass / with the halo pass masked to its 2 rows
set -o pipefail
mkdir -p gpurun_out
L=yolo-v3-tf2_amd/lib
T=tools/tables/bf16_b128_s416_rs.json
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tap_row_reuse" > gpurun_out/r4t_tests.txt 2>&1 || { tail -30 gpurun_out/r4t_tests.txt; exit 1; }
tail -1 gpurun_out/r4t_tests.txt
Y3_LIB_PATH=$PWD/$L/liby3hip_rs_HALO_MASKED.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tap_row_reuse" > gpurun_out/r4t_tests_masked.txt 2>&1 || { tail -30 gpurun_out/r4t_tests_masked.txt; exit 1; }
tail -1 gpurun_out/r4t_tests_masked.txt
timeout -k 10 800 python tools/ab_libs.py $L/liby3hip.so $L/liby3hip.so@$T $L/liby3hip_rs_NO_EXTRA.so@$T $L/liby3hip_rs_HALO_MASKED.so@$T --dtype bf16 --batch 128 --rounds 3 > gpurun_out/r4t_ab_bf16_rs.txt 2>&1 || { tail -20 gpurun_out/r4t_ab_bf16_rs.txt; exit 1; }
grep -v amdgpu gpurun_out/r4t_ab_bf16_rs.txt | tail -4
for v in "" _rs_NO_EXTRA _rs_HALO_MASKED; do
  Y3_LIB_PATH=$PWD/$L/liby3hip$v.so timeout -k 10 300 python tools/tune_tiles.py --dtype bf16 --batch 64 --tiles 24,33 --reps 3 > gpurun_out/r4t_sweep$v.txt 2>&1 || { tail -20 gpurun_out/r4t_sweep$v.txt; exit 1; }
  echo "== liby3hip$v"; grep -v amdgpu gpurun_out/r4t_sweep$v.txt | grep -E "k3s1_c(128|256|512)|sum" | awk '{c[$2]++; if (c[$2] <= 1) print}'
done
