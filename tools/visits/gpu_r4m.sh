#!/bin/bash
# visit 4m: per-queue timeline of one bf16 step (graph replay and eager), to see where only one lane has a kernel running
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
for mode in graph eager; do
  out=gpurun_out/trace_$mode
  mkdir -p $out
  g=""; [ $mode = graph ] && g="--graph"
  timeout -k 10 300 rocprofv3 --kernel-trace -d $out -o run --output-format csv -- python3 bench.py --dtype bf16 --batch 128 $g --steps 6 --warmup 3 --no-cpu-baseline --no-sclk > $out/run.log 2>&1 || { tail -5 $out/run.log; exit 1; }
  f=$(find $out -name "*kernel_trace.csv" | head -1)
  python3 tools/timeline_share.py $f 4 > gpurun_out/4m_share_$mode.txt 2>&1 || { tail gpurun_out/4m_share_$mode.txt; exit 1; }
  python3 tools/timeline_dump.py $f 6 170 > gpurun_out/4m_dump_$mode.txt 2>&1 || { tail gpurun_out/4m_dump_$mode.txt; exit 1; }
  head -1 gpurun_out/4m_share_$mode.txt
  rm -rf $out
done
