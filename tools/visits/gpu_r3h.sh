#!/bin/bash
# round-3 GPU visit H: full parity suite, then rocprofv3 passes (kernel stats + PMC + traffic) for the fp32 headline
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_all.log 2>&1; rc=$?; tail -n 4 gpurun_out/pytest_all.log
[ $rc -ne 0 ] && exit 1
bash tools/profile.sh r03 > gpurun_out/profile_r03.log 2>&1 || { tail -20 gpurun_out/profile_r03.log; exit 1; }
tail -n 40 gpurun_out/profile_r03.log | cut -c1-200
cat gpurun_out/prof_r03/summary_traffic.json
