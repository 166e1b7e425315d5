#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for l in 1 2 3; do
  timeout -k 10 300 python bench.py --dtype bf16 --batch 128 --graph --steps 20 --warmup 5 --no-cpu-baseline --lanes $l > gpurun_out/bench_bf16_l$l.log 2>&1 || { tail -20 gpurun_out/bench_bf16_l$l.log; exit 1; }
  python - <<PY
import json
d=json.loads([x for x in open("gpurun_out/bench_bf16_l$l.log") if x.startswith("{")][-1])
print("lanes $l", d["value"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["sclk_mhz"])
PY
done
bash tools/profile.sh r03_bf16 --dtype bf16 --batch 128 > gpurun_out/profile_r03_bf16.log 2>&1 || { tail -20 gpurun_out/profile_r03_bf16.log; exit 1; }
head -30 gpurun_out/prof_r03_bf16/summary_kernel_stats.txt | cut -c1-130
head -16 gpurun_out/prof_r03_bf16/summary_derived.txt | cut -c1-140
