#!/bin/bash
# visit 4w: the lanes' stem kernels side by side (each takes 1/lanes of the persistent workgroup slots) vs one after the other (Y3_STEM_SHARE=0)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "stem or lane" > gpurun_out/4w_tests.log 2>&1 || { tail -40 gpurun_out/4w_tests.log; exit 1; }
tail -1 gpurun_out/4w_tests.log
for rep in 1 2 3; do
  for sh in 1 0; do
    export Y3_STEM_SHARE=$sh
    timeout -k 10 300 python bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-alt --no-sclk > gpurun_out/4w.log 2>&1 || { tail -20 gpurun_out/4w.log; exit 1; }
    echo "f32 share=$sh rep=$rep $(tail -n 1 gpurun_out/4w.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["roofline"]["frac"])')"
    timeout -k 10 300 python bench.py --dtype bf16 --batch 128 --graph --steps 30 --warmup 10 --no-cpu-baseline --no-sclk > gpurun_out/4w.log 2>&1 || { tail -20 gpurun_out/4w.log; exit 1; }
    echo "bf16 share=$sh rep=$rep $(tail -n 1 gpurun_out/4w.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
  done
done
