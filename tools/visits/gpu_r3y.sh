#!/bin/bash
# visit y: bf16 table A/B: steady-tuned table vs the same with every tile that has a 16x16x32 twin replaced by it
set -o pipefail
mkdir -p gpurun_out
for rep in 1 2 3; do
  for t in tuned m16all; do
    if [ $t = tuned ]; then unset Y3_TUNING_FILE; else export Y3_TUNING_FILE=$PWD/tools/tables/bf16_m16_all.json; fi
    timeout -k 10 300 python bench.py --dtype bf16 --batch 128 --graph --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/y_bench_${t}_$rep.log 2>&1 || { tail -20 gpurun_out/y_bench_${t}_$rep.log; exit 1; }
    echo "$t rep=$rep $(tail -n 1 gpurun_out/y_bench_${t}_$rep.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
  done
done
