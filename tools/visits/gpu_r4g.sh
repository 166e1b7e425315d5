#!/bin/bash
# visit 4g: head conv + decode fused (y3_net_forward_decode): bit-identity tests, then the bench fused vs composed (Y3_FUSE_DECODE=0), fp32 and bf16
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_bench_multirank.py -x -q -k "forward_decode or detect or decode or end_to_end or inference or plugin or hipgraph or comm or bench" > gpurun_out/r4g_tests.log 2>&1 || { tail -60 gpurun_out/r4g_tests.log; exit 1; }
tail -2 gpurun_out/r4g_tests.log
for f in 1 0; do
  Y3_FUSE_DECODE=$f timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-alt --no-cpu-baseline > gpurun_out/r4g_bench_f32_fuse$f.json 2> gpurun_out/r4g_bench_f32_fuse$f.err || { tail -20 gpurun_out/r4g_bench_f32_fuse$f.err; exit 1; }
  Y3_FUSE_DECODE=$f timeout -k 10 400 python bench.py --steps 20 --warmup 5 --dtype bf16 --batch 128 --graph --no-cpu-baseline > gpurun_out/r4g_bench_bf16_fuse$f.json 2> gpurun_out/r4g_bench_bf16_fuse$f.err || { tail -20 gpurun_out/r4g_bench_bf16_fuse$f.err; exit 1; }
done
python - <<'PY'
import json
for n in ("f32_fuse1", "f32_fuse0", "bf16_fuse1", "bf16_fuse0"):
    d = json.load(open(f"gpurun_out/r4g_bench_{n}.json"))
    print(n, d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["ms_per_launch"])
PY
