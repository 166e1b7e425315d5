#!/bin/bash
# visit 4l: start stagger of the first round of workgroups of the 256x256 bf16 tiles (Y3_STAGGER="p,ticks[,lane0only]"; ticks of 10 ns)
set -o pipefail
mkdir -p gpurun_out
for lanes in 1 2; do
  for sg in "0,0" "2,1000" "2,2000" "4,500" "4,1000" "8,250" "8,500" "16,250"; do
    export Y3_STAGGER=$sg
    timeout -k 10 300 python bench.py --dtype bf16 --batch 128 --graph --lanes $lanes --steps 30 --warmup 10 --no-cpu-baseline --no-sclk > gpurun_out/4l_l${lanes}_${sg/,/_}.log 2>&1 || { tail -20 gpurun_out/4l_l${lanes}_${sg/,/_}.log; exit 1; }
    echo "lanes=$lanes stagger=$sg $(tail -n 1 gpurun_out/4l_l${lanes}_${sg/,/_}.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
  done
done
