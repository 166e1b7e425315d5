#!/bin/bash
# round 5, visit 19: the adopted bf16 three-lane table: bf16 GPU tests, the bf16 headline under graph replay (twice), and the 64-image bf16 plan the same way (tuner)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bf16 or lanes" > gpurun_out/r05_19_tests.log 2>&1 || { tail -60 gpurun_out/r05_19_tests.log; exit 1; }
tail -1 gpurun_out/r05_19_tests.log
for k in 1 2; do
  timeout -k 10 600 python bench.py --dtype bf16 --batch 128 --graph --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r05_19_bench_bf16_run$k.json 2> gpurun_out/r05_19_bench.err || { tail -20 gpurun_out/r05_19_bench.err; exit 1; }
  python -c "import json; d = json.load(open('gpurun_out/r05_19_bench_bf16_run$k.json')); print('bench bf16 graph:', d['value'], 'img/s', d['ms_per_step'], 'ms; lanes', d['config']['lanes'], 'frac', d['roofline']['frac'], 'parity', d['parity_checked'])"
done
