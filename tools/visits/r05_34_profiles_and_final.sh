#!/bin/bash
# round 5, visit 34 (repeat of visit 14 on the final library: three-lane bf16 table, round-5 nms_kernel, lane 0 on the caller's stream): the round's rocprofv3 summaries (fp32 headline, bf16 config 5) and the round-end bench lines of every BASELINE config on ONE box
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
bash tools/profile.sh r05 > gpurun_out/r05_34_profile_f32.log 2>&1 || { tail -20 gpurun_out/r05_34_profile_f32.log; exit 1; }
tail -3 gpurun_out/r05_34_profile_f32.log
bash tools/profile.sh r05_bf16 --dtype bf16 --batch 128 > gpurun_out/r05_34_profile_bf16.log 2>&1 || { tail -20 gpurun_out/r05_34_profile_bf16.log; exit 1; }
tail -3 gpurun_out/r05_34_profile_bf16.log
bash tools/gpu_final.sh r05final > gpurun_out/r05_34_final.log 2>&1 || { tail -30 gpurun_out/r05_34_final.log; exit 1; }
grep -E "^== .* rc=" gpurun_out/r05_34_final.log
ls gpurun_out/prof_r05 gpurun_out/prof_r05_bf16 | head -40
