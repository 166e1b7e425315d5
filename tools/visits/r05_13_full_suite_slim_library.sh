#!/bin/bash
# round 5, visit 13: the library with only plan-selectable tiles (2.7 MB instead of 6.1): the whole GPU suite, the fp32 headline and the bf16 config-5 line
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r05_13_tests.log 2>&1 || { tail -60 gpurun_out/r05_13_tests.log; exit 1; }
tail -1 gpurun_out/r05_13_tests.log
timeout -k 10 600 python bench.py > gpurun_out/r05_13_bench_f32.json 2> gpurun_out/r05_13_bench_f32.err || { tail -20 gpurun_out/r05_13_bench_f32.err; exit 1; }
python -c "import json; d = json.load(open('gpurun_out/r05_13_bench_f32.json')); print('f32:', d['value'], 'img/s', d['ms_per_step'], 'ms; frac', d['roofline']['frac'], 'clock-limited', d['roofline']['frac_of_clock_limited_peak'], 'sclk', d['roofline']['sclk_mhz'], 'cpu', d['cpu_baseline']['value'])"
timeout -k 10 600 python bench.py --dtype bf16 --batch 128 --graph --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r05_13_bench_bf16.json 2> gpurun_out/r05_13_bench_bf16.err || { tail -20 gpurun_out/r05_13_bench_bf16.err; exit 1; }
python -c "import json; d = json.load(open('gpurun_out/r05_13_bench_bf16.json')); print('bf16:', d['value'], 'img/s', d['ms_per_step'], 'ms; frac', d['roofline']['frac'])"
