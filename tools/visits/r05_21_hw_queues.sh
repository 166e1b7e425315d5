#!/bin/bash
# round 5, visit 21: GPU_MAX_HW_QUEUES (ROCm maps hipStreams onto 4 hardware queues by default; streams sharing a queue serialize): the bf16 graph step, the
# lanes sweep and the steps-in-flight gate with 4 (default) and 8 / 16 queues, alternating on one box
set -o pipefail
mkdir -p gpurun_out
O=gpurun_out/r05_21_hw_queues.txt
: > $O
for q in 4 8 16 4 8; do
  echo "== GPU_MAX_HW_QUEUES=$q: bench bf16 b128 graph" >> $O
  GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python bench.py --dtype bf16 --batch 128 --graph --steps 30 --warmup 5 --no-cpu-baseline --parity-images 0 --no-sclk > gpurun_out/r05_21_b.json 2> gpurun_out/r05_21_b.err || { tail -20 gpurun_out/r05_21_b.err; exit 1; }
  python -c "import json; d = json.load(open('gpurun_out/r05_21_b.json')); print(d['value'], 'img/s', d['ms_per_step'], 'ms; lanes', d['config']['lanes'])" >> $O
done
for q in 4 8; do
  echo "== GPU_MAX_HW_QUEUES=$q: lanes sweep bf16 b128 (eager conv stack)" >> $O
  GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python tools/lanes_sweep.py --dtype bf16 --batch 128 --lanes 1,2,3,4 >> $O 2> gpurun_out/r05_21_l.err || { tail -20 gpurun_out/r05_21_l.err; exit 1; }
  echo "== GPU_MAX_HW_QUEUES=$q: lanes sweep f32 b64 (eager conv stack)" >> $O
  GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python tools/lanes_sweep.py --dtype f32 --batch 64 --lanes 1,2,3,4 >> $O 2> gpurun_out/r05_21_l.err || { tail -20 gpurun_out/r05_21_l.err; exit 1; }
done
for q in 8 16; do
  echo "== GPU_MAX_HW_QUEUES=$q: steps in flight, bf16 graph" >> $O
  GPU_MAX_HW_QUEUES=$q timeout -k 10 400 python tools/gate_steps_in_flight.py --dtype bf16 --batch 128 --graph --depths 1,2 --offsets 0,0.5 --steps 60 --rounds 1 >> $O 2> gpurun_out/r05_21_g.err || { tail -20 gpurun_out/r05_21_g.err; exit 1; }
  echo "== GPU_MAX_HW_QUEUES=$q: steps in flight, f32 eager" >> $O
  GPU_MAX_HW_QUEUES=$q timeout -k 10 400 python tools/gate_steps_in_flight.py --dtype f32 --batch 64 --depths 1,2 --offsets 0,0.5 --steps 30 --rounds 1 >> $O 2> gpurun_out/r05_21_g.err || { tail -20 gpurun_out/r05_21_g.err; exit 1; }
done
cat $O
