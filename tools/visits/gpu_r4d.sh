#!/bin/bash
# visit 4d: steady-state coordinate descent over the bf16 table (128 images, table's lanes), all current tiles as candidates
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python tools/tune_steady.py --dtype bf16 --batch 128 --steps 25 --write bf16_b128_s416.json > gpurun_out/4d_tune_steady_bf16.txt 2>&1 || { tail -20 gpurun_out/4d_tune_steady_bf16.txt; exit 1; }
tail -15 gpurun_out/4d_tune_steady_bf16.txt
cp yolo-v3-tf2_amd/tuning/bf16_b128_s416.json gpurun_out/4d_bf16_b128_s416.json
