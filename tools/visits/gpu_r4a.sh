#!/bin/bash
# visit 4a (round 4): the whole GPU suite after the round-4 hygiene batch (experimental tiles removed, new tests), then the headline bench
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > gpurun_out/r4a_tests.log 2>&1 || { tail -60 gpurun_out/r4a_tests.log; exit 1; }
tail -3 gpurun_out/r4a_tests.log
grep "stem conv1 output" gpurun_out/r4a_tests.log
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > gpurun_out/r4a_bench_f32.json 2> gpurun_out/r4a_bench_f32.err || { tail -20 gpurun_out/r4a_bench_f32.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4a_bench_f32.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_of_clock_limited_peak"], d["roofline"]["sclk_mhz"], d["parity"]["image_indices"], d["roofline"]["traffic_round"])
PY
