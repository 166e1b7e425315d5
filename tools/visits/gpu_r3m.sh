#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "bf16" > gpurun_out/pytest_bf16.log 2>&1 || { tail -40 gpurun_out/pytest_bf16.log; exit 1; }
tail -n 3 gpurun_out/pytest_bf16.log
timeout -k 10 600 python bench.py --dtype bf16 --batch 128 --graph --steps 20 --warmup 5 --no-cpu-baseline --per-layer > gpurun_out/bench_bf16_epi.log 2>&1 || { tail -30 gpurun_out/bench_bf16_epi.log; exit 1; }
tail -n 1 gpurun_out/bench_bf16_epi.log | cut -c1-300
