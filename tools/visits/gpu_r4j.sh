#!/bin/bash
# visit 4j: bf16 128 x 416^2 bench: table of round 3 / weight-resident kernel on conv3 only / on conv3 + the two 3x3 64->128 @104
set -o pipefail
mkdir -p gpurun_out
for t in base resident_conv3only resident base resident; do
  if [ $t = base ]; then unset Y3_TUNING_FILE; else export Y3_TUNING_FILE=$PWD/tools/tables/bf16_b128_s416_$t.json; fi
  timeout -k 10 400 python bench.py --steps 20 --warmup 5 --dtype bf16 --batch 128 --graph --no-cpu-baseline --parity-images 0 > gpurun_out/r4j_bf16_$t.json 2> gpurun_out/r4j_bf16_$t.err || { tail -20 gpurun_out/r4j_bf16_$t.err; exit 1; }
  python - <<PY
import json
d = json.load(open("gpurun_out/r4j_bf16_$t.json"))
print("$t", d["value"], d["ms_per_step"], d["roofline"]["ms_per_launch"])
PY
done
