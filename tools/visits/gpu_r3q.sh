#!/bin/bash
# visit q: bf16 BK=32 multi-workgroup tiles 21..23: parity, then per-conv sweep against the current winners
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -k "bf16_every_tile" > gpurun_out/q_tests.log 2>&1 || { tail -20 gpurun_out/q_tests.log; exit 1; }
tail -2 gpurun_out/q_tests.log
timeout -k 10 500 python tools/tune_tiles.py --dtype bf16 --batch 128 --reps 3 --tiles 5,10,12,16,17,18,19,21,22,23 > gpurun_out/q_tune.log 2>&1 || { tail -20 gpurun_out/q_tune.log; exit 1; }
tail -90 gpurun_out/q_tune.log
