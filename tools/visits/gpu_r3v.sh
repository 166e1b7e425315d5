#!/bin/bash
# visit v: chunk-major K order as the default: GPU suite, bench A/B against the tap-major order, counter passes
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/v_tests.log 2>&1 || { tail -30 gpurun_out/v_tests.log; exit 1; }
tail -2 gpurun_out/v_tests.log
for rep in 1 2; do
  for ck in 0 -1; do
    if [ $ck = 0 ]; then export Y3_K_CHUNK=0; else unset Y3_K_CHUNK; fi
    timeout -k 10 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/v_bench_ck${ck}_$rep.log 2>&1 || { tail -20 gpurun_out/v_bench_ck${ck}_$rep.log; exit 1; }
    echo "ck=$ck rep=$rep $(tail -n 1 gpurun_out/v_bench_ck${ck}_$rep.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("sclk_mhz"))')"
  done
done
unset Y3_K_CHUNK
bash tools/profile.sh r03b > gpurun_out/v_profile.log 2>&1 || { tail -20 gpurun_out/v_profile.log; exit 1; }
python3 tools/traffic_per_layer.py gpurun_out/prof_r03b > gpurun_out/v_traffic_per_layer.txt 2>&1 || { tail gpurun_out/v_traffic_per_layer.txt; exit 1; }
tail -3 gpurun_out/v_traffic_per_layer.txt
