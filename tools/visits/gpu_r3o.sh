#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/pytest_all.log 2>&1; rc=$?; tail -n 4 gpurun_out/pytest_all.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 600 python bench.py --dtype bf16 --batch 128 --graph --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_bf16_final.log 2>&1 || { tail -30 gpurun_out/bench_bf16_final.log; exit 1; }
tail -n 1 gpurun_out/bench_bf16_final.log | cut -c1-1400
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --per-layer > gpurun_out/bench_f32.log 2>&1 || { tail -30 gpurun_out/bench_f32.log; exit 1; }
tail -n 1 gpurun_out/bench_f32.log | cut -c1-1800
