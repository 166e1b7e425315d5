#!/bin/bash
# visit 4c: staggered lanes (lane l+1 starts when lane l is k ops into the list), bf16 config-5 geometry, graph replay; fp32 graph vs eager
set -o pipefail
mkdir -p gpurun_out
for rep in 1 2; do
  for k in 0 3 4 6 9 14 24; do
    export Y3_LANE_STAGGER=$k
    timeout -k 10 300 python bench.py --dtype bf16 --batch 128 --graph --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/4c_bf16_stag${k}_$rep.log 2>&1 || { tail -20 gpurun_out/4c_bf16_stag${k}_$rep.log; exit 1; }
    echo "stagger=$k rep=$rep $(tail -n 1 gpurun_out/4c_bf16_stag${k}_$rep.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
  done
done
unset Y3_LANE_STAGGER
for g in "" "--graph"; do
  timeout -k 10 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-alt $g > gpurun_out/4c_f32_g.log 2>&1 || { tail -20 gpurun_out/4c_f32_g.log; exit 1; }
  echo "f32 [$g] $(tail -n 1 gpurun_out/4c_f32_g.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["frac"])')"
done
