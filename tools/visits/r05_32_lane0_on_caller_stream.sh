#!/bin/bash
# round 5, visit 32: lane 0 of a multi-lane forward on the caller's stream (no cross-queue signal before its first kernel, the join waits for the other lanes only) against
# lane 0 on a forked stream (Y3_LANE0_OWN_STREAM=1, the form up to here): whole steps, fp32 eager and bf16 graph replay, alternating; then the lanes / detect / graph GPU tests
set -o pipefail
mkdir -p gpurun_out
O=gpurun_out/r05_32_lane0_on_caller_stream.txt
: > $O
for r in 1 2 3; do
  for own in 0 1; do
    Y3_LANE0_OWN_STREAM=$own timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-alt --parity-images 0 --no-sclk > gpurun_out/r05_32_b.json 2> gpurun_out/r05_32.err || { tail -20 gpurun_out/r05_32.err; exit 1; }
    python -c "import json; d = json.load(open('gpurun_out/r05_32_b.json')); print('round $r lane0_forked=$own f32 eager :', d['value'], 'img/s', d['ms_per_step'], 'ms')" >> $O
    Y3_LANE0_OWN_STREAM=$own timeout -k 10 300 python bench.py --dtype bf16 --batch 128 --graph --steps 40 --warmup 5 --no-cpu-baseline --parity-images 0 --no-sclk > gpurun_out/r05_32_b.json 2> gpurun_out/r05_32.err || { tail -20 gpurun_out/r05_32.err; exit 1; }
    python -c "import json; d = json.load(open('gpurun_out/r05_32_b.json')); print('round $r lane0_forked=$own bf16 graph:', d['value'], 'img/s', d['ms_per_step'], 'ms')" >> $O
  done
done
cat $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lanes or detect or graph or full_size or comm" > gpurun_out/r05_32_tests.log 2>&1 || { tail -60 gpurun_out/r05_32_tests.log; exit 1; }
tail -1 gpurun_out/r05_32_tests.log
