#!/bin/bash
# visit 4h: kernel trace of the fused-decode bench (fp32, bf16) -- how long do the head launches take against the composed route?
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
for f in 1 0; do
  export Y3_FUSE_DECODE=$f
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/r4h_f32_fuse$f -o run --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-sclk --parity-images 0 > gpurun_out/r4h_f32_fuse$f.log 2>&1 || { tail -5 gpurun_out/r4h_f32_fuse$f.log; exit 1; }
  python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/r4h_f32_fuse$f/**/run_kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows:
    n = r["Name"]
    if "head" in n or "decode" in n or "nms" in n or ("ELi1ELi" in n and False):
        print("fuse$f", n[:70], r["Calls"], r["AverageNs"])
PY
done
