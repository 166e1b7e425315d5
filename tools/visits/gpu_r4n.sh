#!/bin/bash
# visit 4n: the whole GPU suite on the round-4 library, smoke(), then the default bench line (what the driver runs)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/r4n_tests.log 2>&1 || { tail -60 gpurun_out/r4n_tests.log; exit 1; }
tail -3 gpurun_out/r4n_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4n_smoke.log 2>&1 || { tail -20 gpurun_out/r4n_smoke.log; exit 1; }
tail -1 gpurun_out/r4n_smoke.log
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4n_bench_f32.json 2> gpurun_out/r4n_bench_f32.err || { tail -20 gpurun_out/r4n_bench_f32.err; exit 1; }
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4n_bench_f32.json"))
r = d["roofline"]
print(d["value"], d["ms_per_step"], r["frac"], r["frac_of_clock_limited_peak"], r["sclk_mhz"], r["traffic"], r["traffic_round"], d["parity"]["image_indices"], d["parity"]["max_abs_dbox_raw"], d["cpu_baseline"]["value"])
print({k: v["value"] for k, v in d.items() if k.startswith("alt_")})
PY
