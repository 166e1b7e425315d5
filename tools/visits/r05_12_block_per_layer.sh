#!/bin/bash
# round 5, visit 12: per-conv times (isolated launches, 64 images, one lane) of the early layers with the block fusion on / off
set -o pipefail
mkdir -p gpurun_out
for m in 0 1; do
  echo "== Y3_BLOCK_FUSION=$m"
  Y3_BLOCK_FUSION=$m timeout -k 10 300 python bench.py --dtype bf16 --batch 64 --steps 3 --warmup 2 --no-cpu-baseline --parity-images 0 --no-sclk --per-layer 2>&1 >/dev/null | grep -E "^conv[0-9] |^conv1[0-2] "
done | tee gpurun_out/r05_12_block_per_layer.txt
