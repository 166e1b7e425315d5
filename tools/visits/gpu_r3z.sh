#!/bin/bash
# visit z: new tests (clock timeline, K chunk), default bench with the time-weighted clock
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "sclk or chunk_major or bf16_every_tile" > gpurun_out/z_tests.log 2>&1 || { tail -30 gpurun_out/z_tests.log; exit 1; }
tail -2 gpurun_out/z_tests.log
timeout -k 10 600 python bench.py > gpurun_out/z_bench.json 2> gpurun_out/z_bench.err || { tail -20 gpurun_out/z_bench.err; exit 1; }
python3 -c 'import json; d=json.loads(open("gpurun_out/z_bench.json").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"]); print(json.dumps(d["roofline"], indent=1)); print(d["cpu_baseline"])'
