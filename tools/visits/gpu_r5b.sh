#!/bin/bash
# visit 5b: whole GPU suite (incl. the two-rank rehearsal) and the default bench on HEAD
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/5b_tests.log 2>&1 || { tail -30 gpurun_out/5b_tests.log; exit 1; }
tail -2 gpurun_out/5b_tests.log
timeout -k 10 600 python bench.py > gpurun_out/5b_bench.json 2> gpurun_out/5b_bench.err || { tail -20 gpurun_out/5b_bench.err; exit 1; }
python3 -c 'import json; d=json.loads(open("gpurun_out/5b_bench.json").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_of_clock_limited_peak"], d["cpu_baseline"]["value"], d["parity"]["end_to_end_selection_equal"])'
