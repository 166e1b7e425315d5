#!/bin/bash
# visit 4k: weight-resident bf16 kernel, second build (weights in registers, shortcut requested a tile ahead): tests, isolated sweep, end-to-end A/B
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "weight_resident or bf16" > gpurun_out/r4k_tests.log 2>&1 || { tail -60 gpurun_out/r4k_tests.log; exit 1; }
tail -2 gpurun_out/r4k_tests.log
timeout -k 10 600 python tools/tune_tiles.py --dtype bf16 --batch 128 --tiles 30,22,32 --reps 3 > gpurun_out/r4k_sweep_bf16_resident.txt 2>&1 || { tail -30 gpurun_out/r4k_sweep_bf16_resident.txt; exit 1; }
grep -v amdgpu gpurun_out/r4k_sweep_bf16_resident.txt | awk 'NR==1 || /k3s1_c32|k3s1_c64|sum/'
for t in base resident_conv3only resident base resident; do
  if [ $t = base ]; then unset Y3_TUNING_FILE; else export Y3_TUNING_FILE=$PWD/tools/tables/bf16_b128_s416_$t.json; fi
  timeout -k 10 400 python bench.py --steps 20 --warmup 5 --dtype bf16 --batch 128 --graph --no-cpu-baseline --parity-images 0 > gpurun_out/r4k_bf16_$t.json 2> gpurun_out/r4k_bf16_$t.err || { tail -20 gpurun_out/r4k_bf16_$t.err; exit 1; }
  python - <<PY
import json
d = json.load(open("gpurun_out/r4k_bf16_$t.json"))
print("$t", d["value"], d["ms_per_step"], d["roofline"]["ms_per_launch"])
PY
done
