#!/bin/bash
# visit 4t: steady-state coordinate descent over the fp32 table with TWO lanes (op-major enqueue), then bench lanes 1 vs 2 with the result
set -o pipefail
mkdir -p gpurun_out
cp yolo-v3-tf2_amd/tuning/f32_b64_s416.json gpurun_out/4t_f32_table_before.json
timeout -k 10 900 python tools/tune_steady.py --dtype f32 --batch 64 --lanes 2 --steps 15 --write f32_b64_s416.json > gpurun_out/4t_tune_steady_f32_lanes2.txt 2>&1 || { tail -20 gpurun_out/4t_tune_steady_f32_lanes2.txt; exit 1; }
grep -v "keeps tile" gpurun_out/4t_tune_steady_f32_lanes2.txt | grep -v amdgpu
cp yolo-v3-tf2_amd/tuning/f32_b64_s416.json gpurun_out/4t_f32_table_after.json
for rep in 1 2; do
  for l in 1 2; do
    timeout -k 10 300 python bench.py --lanes $l --steps 30 --warmup 10 --no-cpu-baseline --no-alt --no-sclk > gpurun_out/4t_f32.log 2>&1 || { tail -20 gpurun_out/4t_f32.log; exit 1; }
    echo "tuned-for-2-lanes table, lanes=$l rep=$rep $(tail -n 1 gpurun_out/4t_f32.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["frac"])')"
  done
done
