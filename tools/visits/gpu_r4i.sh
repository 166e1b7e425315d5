#!/bin/bash
# visit 4i: bf16 tiles 30 / 31 (LDS-DMA, BK 32, 64 output channels): tile tests, per-conv sweep on the Cin = 32 / small layers, steady A/B
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "bf16_every_tile or teacher_forced" > gpurun_out/4i_tests.log 2>&1 || { tail -40 gpurun_out/4i_tests.log; exit 1; }
tail -2 gpurun_out/4i_tests.log
timeout -k 10 500 python tools/tune_tiles.py --dtype bf16 --batch 64 --reps 3 --tiles 5,30,31,6,22,10,2 > gpurun_out/4i_sweep.txt 2>&1 || { tail -20 gpurun_out/4i_sweep.txt; exit 1; }
grep -E "^(conv|1 |3 |4 |5 |6 |7 |8 ) " gpurun_out/4i_sweep.txt | cut -c1-160
