#!/bin/bash
# visit 5m: steady-state coordinate descent over the fp32 table again (two lanes) after the prologue / epilogue / register changes
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python tools/tune_steady.py --dtype f32 --batch 64 --steps 15 --write f32_b64_s416.json > gpurun_out/5m_tune_steady_f32.txt 2>&1 || { tail -20 gpurun_out/5m_tune_steady_f32.txt; exit 1; }
grep -v "keeps tile" gpurun_out/5m_tune_steady_f32.txt | grep -v amdgpu
cp yolo-v3-tf2_amd/tuning/f32_b64_s416.json gpurun_out/5m_f32_b64_s416.json
