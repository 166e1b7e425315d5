#!/bin/bash
# visit t: bf16 tiles on 16x16x32 MFMAs (24..29): parity, isolated sweep, steady re-tune, bench
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -k "bf16_every_tile" > gpurun_out/t_tests.log 2>&1 || { tail -20 gpurun_out/t_tests.log; exit 1; }
tail -2 gpurun_out/t_tests.log
timeout -k 10 500 python tools/tune_tiles.py --dtype bf16 --batch 128 --reps 3 --tiles 8,10,12,17,18,19,24,25,26,27,28,29 > gpurun_out/t_tune.log 2>&1 || { tail -20 gpurun_out/t_tune.log; exit 1; }
grep -v Warning gpurun_out/t_tune.log | cut -c1-190 | tail -80
timeout -k 10 900 python tools/tune_steady.py --dtype bf16 --batch 128 --steps 20 --write bf16_b128_s416.json > gpurun_out/t_steady.log 2>&1 || { tail -20 gpurun_out/t_steady.log; exit 1; }
grep -- "->\|start\|final" gpurun_out/t_steady.log
cp yolo-v3-tf2_amd/tuning/bf16_b128_s416.json gpurun_out/t_bf16_b128_s416.json
timeout -k 10 300 python bench.py --dtype bf16 --batch 128 --graph --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/t_bench_after.log 2>&1 || { tail -20 gpurun_out/t_bench_after.log; exit 1; }
tail -n 1 gpurun_out/t_bench_after.log | cut -c1-300
