#!/bin/bash
# visit 4g: bf16 stem, third layer inside the kernel (mode 1) vs as its own launch (mode 2), same box, interleaved
set -o pipefail
mkdir -p gpurun_out
for rep in 1 2 3; do
  for m in 1 2; do
    export Y3_STEM_MODE=$m
    timeout -k 10 300 python bench.py --dtype bf16 --batch 128 --graph --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/4g_bf16_m${m}_$rep.log 2>&1 || { tail -20 gpurun_out/4g_bf16_m${m}_$rep.log; exit 1; }
    echo "mode=$m rep=$rep $(tail -n 1 gpurun_out/4g_bf16_m${m}_$rep.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
  done
done
