#!/bin/bash
# visit 4y: smoke() + the 608 tests + bench at 608 with the re-tuned table
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/4y_smoke.log 2>&1 || { tail -20 gpurun_out/4y_smoke.log; exit 1; }
tail -1 gpurun_out/4y_smoke.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "608" > gpurun_out/4y_tests.log 2>&1 || { tail -40 gpurun_out/4y_tests.log; exit 1; }
tail -1 gpurun_out/4y_tests.log
timeout -k 10 300 python bench.py --image-size 608 --steps 10 --warmup 3 --no-alt --no-cpu-baseline > gpurun_out/r03_bench_f32_b64_s608.json 2> gpurun_out/4y.err || { tail -20 gpurun_out/4y.err; exit 1; }
python3 -c 'import json; d=json.loads(open("gpurun_out/r03_bench_f32_b64_s608.json").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["parity"]["end_to_end_selection_equal"])'
