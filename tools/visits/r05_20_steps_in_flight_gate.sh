#!/bin/bash
# round 5, visit 20: gate -- two / three batches in flight (one net + stream per batch, consecutive steps overlap) against the serial step, bf16 graph replay and fp32 eager
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python tools/gate_steps_in_flight.py --dtype bf16 --batch 128 --graph --depths 1,2,3 --offsets 0,0.5 --steps 60 > gpurun_out/r05_20_gate_bf16.txt 2> gpurun_out/r05_20_gate_bf16.err || { tail -30 gpurun_out/r05_20_gate_bf16.err; exit 1; }
cat gpurun_out/r05_20_gate_bf16.txt
timeout -k 10 500 python tools/gate_steps_in_flight.py --dtype f32 --batch 64 --depths 1,2 --offsets 0,0.5 --steps 30 > gpurun_out/r05_20_gate_f32.txt 2> gpurun_out/r05_20_gate_f32.err || { tail -30 gpurun_out/r05_20_gate_f32.err; exit 1; }
cat gpurun_out/r05_20_gate_f32.txt
