#!/bin/bash
# visit 5d: bf16 bench line with its parity block (head logits vs the bf16 oracle, floor measured in the run)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python bench.py --dtype bf16 --batch 128 --graph --no-cpu-baseline > gpurun_out/5d_bf16.json 2> gpurun_out/5d_bf16.err || { tail -20 gpurun_out/5d_bf16.err; exit 1; }
python3 -c 'import json; d=json.loads(open("gpurun_out/5d_bf16.json").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["parity_checked"]); print(json.dumps(d["parity"], indent=1))'
