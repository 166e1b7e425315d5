#!/bin/bash
# visit 4v: rocprofv3 passes of the fp32 headline as it ships now (two lanes) and of the bf16 geometry; kernel-trace timelines shared out
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
bash tools/profile.sh r03f32 > gpurun_out/4v_profile_f32.log 2>&1 || { tail -20 gpurun_out/4v_profile_f32.log; exit 1; }
f=$(find gpurun_out/prof_r03f32/stats -name "*kernel_trace.csv" | head -1)
python3 tools/timeline_share.py $f 2 > gpurun_out/4v_timeline_f32.txt 2>&1 || tail -3 gpurun_out/4v_timeline_f32.txt
python3 tools/traffic_per_layer.py gpurun_out/prof_r03f32 > gpurun_out/4v_traffic_per_layer_f32.txt 2>&1 || tail -5 gpurun_out/4v_traffic_per_layer_f32.txt
bash tools/profile.sh r03bf16 --dtype bf16 --batch 128 > gpurun_out/4v_profile_bf16.log 2>&1 || { tail -20 gpurun_out/4v_profile_bf16.log; exit 1; }
f=$(find gpurun_out/prof_r03bf16/stats -name "*kernel_trace.csv" | head -1)
python3 tools/timeline_share.py $f 2 > gpurun_out/4v_timeline_bf16.txt 2>&1 || tail -3 gpurun_out/4v_timeline_bf16.txt
find gpurun_out/prof_r03f32 gpurun_out/prof_r03bf16 -name "*.csv" -size +20M -delete
find gpurun_out/prof_r03f32 gpurun_out/prof_r03bf16 -name "*.db" -delete
tail -3 gpurun_out/prof_r03f32/summary_kernel_stats.txt; head -12 gpurun_out/4v_timeline_f32.txt; tail -2 gpurun_out/4v_traffic_per_layer_f32.txt
tail -2 gpurun_out/prof_r03bf16/summary_kernel_stats.txt; head -3 gpurun_out/4v_timeline_bf16.txt
cat gpurun_out/prof_r03f32/summary_traffic.json gpurun_out/prof_r03bf16/summary_traffic.json
