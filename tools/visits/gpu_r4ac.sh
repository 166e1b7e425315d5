#!/bin/bash
# visit 4ac: bf16 tile 36 (256x256 on four waves of 128x128, hand-pipelined, conv_bf16_w4.hip): parity against tile 17 bit for bit, then
# the isolated sweep against tiles 17 / 24
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "four_wave" > gpurun_out/r4ac_tests.txt 2>&1 || { tail -30 gpurun_out/r4ac_tests.txt; exit 1; }
tail -1 gpurun_out/r4ac_tests.txt
for b in 64 128; do
timeout -k 10 600 python tools/tune_tiles.py --dtype bf16 --batch $b --tiles 17,24,36 --reps 3 > gpurun_out/r4ac_sweep_b$b.txt 2>&1 || { tail -20 gpurun_out/r4ac_sweep_b$b.txt; exit 1; }
echo "== batch $b"; grep -v amdgpu gpurun_out/r4ac_sweep_b$b.txt | grep -E "k3s[12]_c(128|256|512)|k1s1_c1024_n512|conv  shape|sum" | awk '{c[$2]++; if (c[$2] <= 1 || $1 == "conv") print}'
done
