#!/bin/bash
# visit 4j: bf16 config-5 geometry, eager launches on two lane streams vs the same step replayed from a HIP graph; lanes 2 vs 3
set -o pipefail
mkdir -p gpurun_out
for rep in 1 2; do
  for mode in "--graph" "" "--graph --lanes 3" "--lanes 3" "--graph --lanes 1"; do
    tag=$(echo "$mode" | tr -d ' -')
    timeout -k 10 300 python bench.py --dtype bf16 --batch 128 $mode --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/4j_bf16_${tag}_$rep.log 2>&1 || { tail -20 gpurun_out/4j_bf16_${tag}_$rep.log; exit 1; }
    echo "[$mode] rep=$rep $(tail -n 1 gpurun_out/4j_bf16_${tag}_$rep.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["ms_per_launch"])')"
  done
done
