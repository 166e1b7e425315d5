#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python bench.py --dtype bf16 --batch 128 --graph --steps 20 --warmup 5 --no-cpu-baseline --per-layer > gpurun_out/bench_bf16.log 2>&1 || { tail -30 gpurun_out/bench_bf16.log; exit 1; }
tail -n 80 gpurun_out/bench_bf16.log | cut -c1-200
