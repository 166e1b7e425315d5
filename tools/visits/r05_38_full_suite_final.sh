#!/bin/bash
# round 5, visit 38: the whole GPU suite, smoke() and the default bench line on the final library of the round (round-5 nms_kernel, lane 0 on the caller's stream, three-lane bf16 table)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r05_38_tests.log 2>&1 || { tail -60 gpurun_out/r05_38_tests.log; exit 1; }
tail -1 gpurun_out/r05_38_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_38_smoke.log 2>&1 || { tail -20 gpurun_out/r05_38_smoke.log; exit 1; }
tail -1 gpurun_out/r05_38_smoke.log
timeout -k 10 600 python bench.py > gpurun_out/r05_38_bench_f32.json 2> gpurun_out/r05_38_bench_f32.err || { tail -20 gpurun_out/r05_38_bench_f32.err; exit 1; }
python -c "import json; d = json.load(open('gpurun_out/r05_38_bench_f32.json')); print('f32:', d['value'], 'img/s', d['ms_per_step'], 'ms; frac', d['roofline']['frac'], 'clock-limited', d['roofline']['frac_of_clock_limited_peak'], 'sclk', d['roofline']['sclk_mhz'], 'cpu', d['cpu_baseline']['value'])"
