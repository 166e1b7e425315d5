#!/bin/bash
# visit w: mask-based per-tap address update: GPU suite, bench A/B of the K orders
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/w_tests.log 2>&1 || { tail -30 gpurun_out/w_tests.log; exit 1; }
tail -2 gpurun_out/w_tests.log
for rep in 1 2 3; do
  for ck in 0 -1 128; do
    if [ $ck = -1 ]; then unset Y3_K_CHUNK; else export Y3_K_CHUNK=$ck; fi
    timeout -k 10 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/w_bench_ck${ck}_$rep.log 2>&1 || { tail -20 gpurun_out/w_bench_ck${ck}_$rep.log; exit 1; }
    echo "ck=$ck rep=$rep $(tail -n 1 gpurun_out/w_bench_ck${ck}_$rep.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("sclk_mhz"))')"
  done
done
