#!/bin/bash
# visit 4ad: tile 36 with the shortcut rows of all four blocks requested at the top of the epilogue: parity, sweep against 24
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "four_wave" > gpurun_out/r4ad_tests.txt 2>&1 || { tail -30 gpurun_out/r4ad_tests.txt; exit 1; }
tail -1 gpurun_out/r4ad_tests.txt
timeout -k 10 600 python tools/tune_tiles.py --dtype bf16 --batch 64 --tiles 24,36 --reps 3 > gpurun_out/r4ad_sweep_b64.txt 2>&1 || { tail -20 gpurun_out/r4ad_sweep_b64.txt; exit 1; }
grep -v amdgpu gpurun_out/r4ad_sweep_b64.txt | grep -E "k3s[12]_c(128|256|512)|k1s1_c1024_n512|conv  shape|sum" | awk '{c[$2]++; if (c[$2] <= 1 || $1 == "conv") print}'
