#!/bin/bash
# round 5, visit 35: how many conv kernels run at once over a step, and which own each tenth of it (rocprofv3 --kernel-trace): bf16 graph replay (3 lanes), fp32 eager (2 lanes)
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out/r05_35_concurrency_profile.txt
: > $O
rm -rf gpurun_out/prof_r05_35a gpurun_out/prof_r05_35b
timeout -k 10 400 rocprofv3 --kernel-trace -d gpurun_out/prof_r05_35a -o run --output-format csv -- python3 bench.py --dtype bf16 --batch 128 --graph --steps 8 --warmup 2 --no-cpu-baseline --parity-images 0 --no-sclk > gpurun_out/r05_35a.log 2>&1 || { tail -20 gpurun_out/r05_35a.log; exit 1; }
echo "== bf16 128 x 416^2, graph replay, 3 lanes" >> $O
python3 tools/concurrency_profile.py $(find gpurun_out/prof_r05_35a -name '*kernel_trace.csv' | head -1) 5 >> $O
timeout -k 10 400 rocprofv3 --kernel-trace -d gpurun_out/prof_r05_35b -o run --output-format csv -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-alt --parity-images 0 --no-sclk > gpurun_out/r05_35b.log 2>&1 || { tail -20 gpurun_out/r05_35b.log; exit 1; }
echo "== f32 64 x 416^2, eager, 2 lanes" >> $O
python3 tools/concurrency_profile.py $(find gpurun_out/prof_r05_35b -name '*kernel_trace.csv' | head -1) 4 >> $O
cat $O
rm -rf gpurun_out/prof_r05_35a gpurun_out/prof_r05_35b
