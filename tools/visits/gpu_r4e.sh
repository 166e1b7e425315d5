#!/bin/bash
# visit 4e: kernel-trace timeline of the bf16 step (2 lanes, graph replay): attributed time per kernel and grid
set -o pipefail
export TMPDIR=/tmp
out=gpurun_out/trace_bf16
mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace -d $out -o run --output-format csv -- python3 bench.py --dtype bf16 --batch 128 --graph --steps 8 --warmup 3 --no-cpu-baseline > $out/run.log 2>&1 || { tail -5 $out/run.log; exit 1; }
f=$(find $out -name "*kernel_trace.csv" | head -1)
python3 tools/timeline_share.py $f 4 > gpurun_out/4e_timeline_bf16.txt 2>&1 || { tail gpurun_out/4e_timeline_bf16.txt; exit 1; }
head -40 gpurun_out/4e_timeline_bf16.txt
rm -rf $out
