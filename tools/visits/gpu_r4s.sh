#!/bin/bash
# visit 4s: fp32 headline with 1 / 2 / 3 lanes under the op-major enqueue (the r01 sweep enqueued lane by lane), eager and graph replay
set -o pipefail
mkdir -p gpurun_out
for rep in 1 2; do
  for mode in "--lanes 1" "--lanes 2" "--lanes 2 --graph" "--lanes 3" "--lanes 3 --graph"; do
    timeout -k 10 300 python bench.py $mode --steps 30 --warmup 10 --no-cpu-baseline --no-alt --no-sclk > gpurun_out/4s_f32.log 2>&1 || { tail -20 gpurun_out/4s_f32.log; exit 1; }
    echo "f32 [$mode] rep=$rep $(tail -n 1 gpurun_out/4s_f32.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["frac"])')"
  done
done
