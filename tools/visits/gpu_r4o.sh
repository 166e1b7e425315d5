#!/bin/bash
# visit 4o: round-4 measurement set on the final library: profiles (kernel stats + PMC) of the fp32 headline and of bf16 config 5,
# per-layer clock / rate table, and the BASELINE configs through bench.py (tools/gpu_final.sh)
set -o pipefail
mkdir -p gpurun_out
bash tools/profile.sh r04 > gpurun_out/r4o_profile_r04.log 2>&1 || { tail -20 gpurun_out/r4o_profile_r04.log; exit 1; }
echo "profile f32 done"
bash tools/profile.sh r04_bf16 --dtype bf16 --batch 128 > gpurun_out/r4o_profile_r04_bf16.log 2>&1 || { tail -20 gpurun_out/r4o_profile_r04_bf16.log; exit 1; }
echo "profile bf16 done"
timeout -k 10 400 python tools/sclk_per_layer.py > gpurun_out/r4o_sclk_per_layer_f32_b64_s416.txt 2>&1 || { tail -20 gpurun_out/r4o_sclk_per_layer_f32_b64_s416.txt; exit 1; }
tail -3 gpurun_out/r4o_sclk_per_layer_f32_b64_s416.txt
bash tools/gpu_final.sh r04 > gpurun_out/r4o_final.log 2>&1; grep "== " gpurun_out/r4o_final.log | head -20
