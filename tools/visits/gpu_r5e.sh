#!/bin/bash
# visit 5e: default bench line (config.lanes, kernel string), rehearsal test again
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python bench.py > gpurun_out/r03_bench_f32_b64_s416.json 2> gpurun_out/5e.err || { tail -20 gpurun_out/5e.err; exit 1; }
python3 -c 'import json; d=json.loads(open("gpurun_out/r03_bench_f32_b64_s416.json").read().strip().splitlines()[-1]); print(d["value"], d["roofline"]["frac"], d["config"]["lanes"], d["roofline"]["kernel"], d["alt_f32x2"]["value"])'
timeout -k 10 600 python -m pytest tests/test_bench_multirank.py -x -q -m gpu 2>&1 | tail -1
