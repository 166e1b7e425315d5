#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fused_stem" > gpurun_out/pytest_stem.log 2>&1 || { tail -40 gpurun_out/pytest_stem.log; exit 1; }
tail -n 3 gpurun_out/pytest_stem.log
timeout -k 10 600 python bench.py --dtype bf16 --batch 128 --graph --steps 20 --warmup 5 --no-cpu-baseline --per-layer > gpurun_out/bench_bf16_stem.log 2>&1 || { tail -30 gpurun_out/bench_bf16_stem.log; exit 1; }
head -12 gpurun_out/bench_bf16_stem.log | cut -c1-120; tail -n 1 gpurun_out/bench_bf16_stem.log | cut -c1-400
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_all.log 2>&1; tail -n 5 gpurun_out/pytest_all.log
