#!/bin/bash
# visit 4b: bf16 per-conv tile table at the lane size (64 images) and at 128, isolated launches
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python tools/tune_tiles.py --dtype bf16 --batch 64 --reps 3 --tiles 17,24,18,25,19,26,21,22,23,8,27,12,29,13,10,28 > gpurun_out/4b_sweep_bf16_b64.txt 2>&1 || { tail -20 gpurun_out/4b_sweep_bf16_b64.txt; exit 1; }
tail -3 gpurun_out/4b_sweep_bf16_b64.txt
timeout -k 10 500 python tools/tune_tiles.py --dtype bf16 --batch 128 --reps 3 --tiles 17,24,18,25,21,22,8,27 > gpurun_out/4b_sweep_bf16_b128.txt 2>&1 || { tail -20 gpurun_out/4b_sweep_bf16_b128.txt; exit 1; }
tail -3 gpurun_out/4b_sweep_bf16_b128.txt
