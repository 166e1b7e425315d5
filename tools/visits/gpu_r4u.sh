#!/bin/bash
# visit 4u: lanes 1 vs 2 for the other fp32 geometries (64 x 608^2, backbone-only 32 x 416^2 via bench --batch 32), default bench with the new table
set -o pipefail
mkdir -p gpurun_out
for rep in 1 2; do
  for l in 1 2; do
    timeout -k 10 300 python bench.py --image-size 608 --lanes $l --steps 15 --warmup 5 --no-cpu-baseline --no-alt --no-sclk > gpurun_out/4u.log 2>&1 || { tail -20 gpurun_out/4u.log; exit 1; }
    echo "608 lanes=$l rep=$rep $(tail -n 1 gpurun_out/4u.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["frac"])')"
    timeout -k 10 300 python bench.py --batch 32 --lanes $l --steps 30 --warmup 10 --no-cpu-baseline --no-alt --no-sclk > gpurun_out/4u.log 2>&1 || { tail -20 gpurun_out/4u.log; exit 1; }
    echo "b32 lanes=$l rep=$rep $(tail -n 1 gpurun_out/4u.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["frac"])')"
  done
done
timeout -k 10 500 python bench.py > gpurun_out/4u_bench.json 2> gpurun_out/4u_bench.err || { tail -20 gpurun_out/4u_bench.err; exit 1; }
python3 -c 'import json; d=json.loads(open("gpurun_out/4u_bench.json").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["sclk_mhz"], d["roofline"]["frac_of_clock_limited_peak"], d["parity"]["end_to_end_selection_equal"])'
