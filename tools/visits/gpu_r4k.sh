#!/bin/bash
# visit 4k: rocprofv3 passes (kernel stats + five PMC passes) of the bf16 config-5 geometry on HEAD
set -o pipefail
mkdir -p gpurun_out
bash tools/profile.sh r03bf16 --dtype bf16 --batch 128 > gpurun_out/4k_profile.log 2>&1 || { tail -20 gpurun_out/4k_profile.log; exit 1; }
tail -5 gpurun_out/4k_profile.log
ls gpurun_out/prof_r03bf16/
find gpurun_out/prof_r03bf16 -name "*.csv" -size +20M -delete
find gpurun_out/prof_r03bf16 -name "*.db" -delete
du -sh gpurun_out/prof_r03bf16
bash tools/profile.sh r03f32 > gpurun_out/4k_profile_f32.log 2>&1 || { tail -20 gpurun_out/4k_profile_f32.log; exit 1; }
find gpurun_out/prof_r03f32 -name "*.csv" -size +20M -delete
find gpurun_out/prof_r03f32 -name "*.db" -delete
python3 tools/traffic_per_layer.py gpurun_out/prof_r03f32 > gpurun_out/4k_traffic_per_layer_f32.txt 2>&1 || tail -5 gpurun_out/4k_traffic_per_layer_f32.txt
tail -2 gpurun_out/4k_traffic_per_layer_f32.txt
cat gpurun_out/prof_r03f32/summary_traffic.json gpurun_out/prof_r03bf16/summary_traffic.json
