#!/bin/bash
# visit 4q: tap-row reuse (bf16 tiles 33..35, conv_bf16_rs.hip): parity first, then the isolated per-layer sweep against the tiles of
# the same shapes (24, 27, 29) at the per-lane batch (64) and the whole batch (128)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tap_row_reuse or weight_resident" > gpurun_out/r4q_tests.txt 2>&1 || { tail -30 gpurun_out/r4q_tests.txt; exit 1; }
tail -3 gpurun_out/r4q_tests.txt
for b in 64 128; do
  timeout -k 10 400 python tools/tune_tiles.py --dtype bf16 --batch $b --tiles 24,27,29,33,34,35 --reps 3 > gpurun_out/r4q_sweep_b$b.txt 2>&1 || { tail -20 gpurun_out/r4q_sweep_b$b.txt; exit 1; }
  echo "== batch $b"; grep -v amdgpu gpurun_out/r4q_sweep_b$b.txt | grep -E "k3s1_c(128|256|512)|conv  shape|sum" | awk '{c[$2]++; if (c[$2] <= 1 || $1 == "conv") print}'
done
