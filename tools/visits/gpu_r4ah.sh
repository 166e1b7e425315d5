#!/bin/bash
# visit 4ah: the two selectable bf16 tile families at the real layer shapes: bench.py --dtype bf16 --batch 128 with its parity gate (first image's
# head logits against the bf16-emulating oracle, bounded by the oracle-vs-oracle floor) under a table that puts them on every signature they fit
set -o pipefail
mkdir -p gpurun_out
for t in rs w4; do
  Y3_TUNING_FILE=$PWD/tools/tables/bf16_b128_s416_$t.json timeout -k 10 400 python bench.py --dtype bf16 --batch 128 --graph --steps 10 --warmup 3 --no-cpu-baseline --no-alt > gpurun_out/r4ah_bench_bf16_$t.json 2> gpurun_out/r4ah_bench_bf16_$t.err || { tail -20 gpurun_out/r4ah_bench_bf16_$t.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4ah_bench_bf16_$t.json").read().strip().splitlines()[-1])
print("$t", d["value"], d["ms_per_step"], {k:v for k,v in d.get("parity",{}).items() if k in ("parity_checked","rel_l2","floor_rel_l2","bar")})
PY
done
