#!/bin/bash
# visit 4n: op-major enqueue of the lanes: lane tests, bf16 eager vs graph, default fp32 bench (alt_f32x2 / alt_f32x3 run eager with 3 lanes)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "lane or bf16_full or graph or early_chunk" > gpurun_out/4n_tests.log 2>&1 || { tail -40 gpurun_out/4n_tests.log; exit 1; }
tail -2 gpurun_out/4n_tests.log
for rep in 1 2; do
  for mode in "--graph" ""; do
    timeout -k 10 300 python bench.py --dtype bf16 --batch 128 $mode --steps 30 --warmup 10 --no-cpu-baseline --no-sclk > gpurun_out/4n_bf16.log 2>&1 || { tail -20 gpurun_out/4n_bf16.log; exit 1; }
    echo "bf16 [$mode] rep=$rep $(tail -n 1 gpurun_out/4n_bf16.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
  done
done
timeout -k 10 500 python bench.py --no-cpu-baseline > gpurun_out/4n_f32.json 2> gpurun_out/4n_f32.err || { tail -20 gpurun_out/4n_f32.err; exit 1; }
python3 -c 'import json; d=json.loads(open("gpurun_out/4n_f32.json").read().strip().splitlines()[-1]); print(d["value"], d["roofline"]["frac"], d["alt_f32x3"]["value"], d["alt_f32x2"]["value"])'
