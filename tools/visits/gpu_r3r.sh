#!/bin/bash
# visit r: steady-state re-tune of the bf16 table with the BK=32 multi-workgroup tiles, then the bf16 bench before/after
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --dtype bf16 --batch 128 --graph --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r_bench_before.log 2>&1 || { tail -20 gpurun_out/r_bench_before.log; exit 1; }
tail -n 1 gpurun_out/r_bench_before.log | cut -c1-300
timeout -k 10 600 python tools/tune_steady.py --dtype bf16 --batch 128 --steps 20 --write bf16_b128_s416.json > gpurun_out/r_tune.log 2>&1 || { tail -20 gpurun_out/r_tune.log; exit 1; }
grep -- "->\|start\|final" gpurun_out/r_tune.log
cp yolo-v3-tf2_amd/tuning/bf16_b128_s416.json gpurun_out/r_bf16_b128_s416.json
timeout -k 10 300 python bench.py --dtype bf16 --batch 128 --graph --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r_bench_after.log 2>&1 || { tail -20 gpurun_out/r_bench_after.log; exit 1; }
tail -n 1 gpurun_out/r_bench_after.log | cut -c1-300
