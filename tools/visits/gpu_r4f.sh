#!/bin/bash
# visit 4f: bf16 stem with the 1x1 third layer (phase 3): tests, then bench A/B (Y3_STEM_MODE env is read by nothing: A/B through git stash is not possible on the box, so bench both modes via tools/ab_stem_mode.py)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "stem or bf16" > gpurun_out/4f_tests.log 2>&1 || { tail -40 gpurun_out/4f_tests.log; exit 1; }
tail -2 gpurun_out/4f_tests.log
for rep in 1 2; do
  timeout -k 10 300 python bench.py --dtype bf16 --batch 128 --graph --steps 30 --warmup 10 --no-cpu-baseline > gpurun_out/4f_bf16_$rep.log 2>&1 || { tail -20 gpurun_out/4f_bf16_$rep.log; exit 1; }
  echo "rep=$rep $(tail -n 1 gpurun_out/4f_bf16_$rep.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
done
