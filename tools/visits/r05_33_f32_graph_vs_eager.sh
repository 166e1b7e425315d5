#!/bin/bash
# round 5, visit 33: the fp32 headline step eager against graph replay (bench.py --graph), alternating, on the library with lane 0 on the caller's stream
set -o pipefail
mkdir -p gpurun_out
O=gpurun_out/r05_33_f32_graph_vs_eager.txt
: > $O
for r in 1 2 3; do
  for g in "" "--graph"; do
    timeout -k 10 300 python bench.py $g --steps 30 --warmup 5 --no-cpu-baseline --no-alt --parity-images 0 --no-sclk > gpurun_out/r05_33_b.json 2> gpurun_out/r05_33.err || { tail -20 gpurun_out/r05_33.err; exit 1; }
    python -c "import json; d = json.load(open('gpurun_out/r05_33_b.json')); print('round $r f32 ${g:-eager}:', d['value'], 'img/s', d['ms_per_step'], 'ms; conv stack', d['roofline']['ms_per_launch'])" >> $O
  done
done
cat $O
