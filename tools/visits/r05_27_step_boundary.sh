#!/bin/bash
# round 5, visit 27: the boundary between two steps on the chip's own timeline (rocprofv3 --kernel-trace): bf16 graph replay and fp32 eager
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out/r05_27_step_boundary.txt
: > $O
rm -rf gpurun_out/prof_r05_27a gpurun_out/prof_r05_27b
timeout -k 10 400 rocprofv3 --kernel-trace -d gpurun_out/prof_r05_27a -o run --output-format csv -- python3 bench.py --dtype bf16 --batch 128 --graph --steps 8 --warmup 2 --no-cpu-baseline --parity-images 0 --no-sclk > gpurun_out/r05_27a.log 2>&1 || { tail -20 gpurun_out/r05_27a.log; exit 1; }
echo "== bf16 128 x 416^2, graph replay" >> $O
python3 tools/step_boundary.py $(find gpurun_out/prof_r05_27a -name '*kernel_trace.csv' | head -1) 4 >> $O
timeout -k 10 400 rocprofv3 --kernel-trace -d gpurun_out/prof_r05_27b -o run --output-format csv -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-alt --parity-images 0 --no-sclk > gpurun_out/r05_27b.log 2>&1 || { tail -20 gpurun_out/r05_27b.log; exit 1; }
echo "== f32 64 x 416^2, eager" >> $O
python3 tools/step_boundary.py $(find gpurun_out/prof_r05_27b -name '*kernel_trace.csv' | head -1) 4 >> $O
cat $O
rm -rf gpurun_out/prof_r05_27a gpurun_out/prof_r05_27b
