#!/bin/bash
# visit 4l: round-4 profiles of the headline (fp32 64 x 416^2) and of 64 x 608^2: kernel stats + PMC passes (tools/profile.sh)
set -o pipefail
mkdir -p gpurun_out
bash tools/profile.sh r04 > gpurun_out/r4l_profile_r04.log 2>&1 || { tail -20 gpurun_out/r4l_profile_r04.log; exit 1; }
tail -5 gpurun_out/r4l_profile_r04.log
bash tools/profile.sh r04s608 --image-size 608 > gpurun_out/r4l_profile_r04s608.log 2>&1 || { tail -20 gpurun_out/r4l_profile_r04s608.log; exit 1; }
tail -5 gpurun_out/r4l_profile_r04s608.log
ls gpurun_out/prof_r04 gpurun_out/prof_r04s608 | head -30
