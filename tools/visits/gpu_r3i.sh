#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sclk or fused_stem" > gpurun_out/pytest_sclk.log 2>&1 || { tail -30 gpurun_out/pytest_sclk.log; exit 1; }
tail -n 2 gpurun_out/pytest_sclk.log
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --per-layer > gpurun_out/bench_f32.log 2>&1 || { tail -30 gpurun_out/bench_f32.log; exit 1; }
tail -n 1 gpurun_out/bench_f32.log | cut -c1-1500
timeout -k 10 600 python bench.py --steps 10 --warmup 3 --image-size 608 --no-alt > gpurun_out/bench_f32_608.log 2>&1 || { tail -30 gpurun_out/bench_f32_608.log; exit 1; }
tail -n 1 gpurun_out/bench_f32_608.log | cut -c1-1200
