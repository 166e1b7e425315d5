#!/bin/bash
# visit 4a (session 2 of round 3): whole GPU suite on HEAD, default bench, bf16 config-5 bench
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/4a_tests.log 2>&1 || { tail -30 gpurun_out/4a_tests.log; exit 1; }
tail -2 gpurun_out/4a_tests.log
timeout -k 10 600 python bench.py > gpurun_out/4a_bench.json 2> gpurun_out/4a_bench.err || { tail -20 gpurun_out/4a_bench.err; exit 1; }
python3 -c 'import json; d=json.loads(open("gpurun_out/4a_bench.json").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"]); print(json.dumps(d["roofline"])); print(d["cpu_baseline"])'
timeout -k 10 300 python bench.py --dtype bf16 --batch 128 --graph --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/4a_bench_bf16.json 2> gpurun_out/4a_bench_bf16.err || { tail -20 gpurun_out/4a_bench_bf16.err; exit 1; }
python3 -c 'import json; d=json.loads(open("gpurun_out/4a_bench_bf16.json").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"]); print(json.dumps(d["roofline"]))'
