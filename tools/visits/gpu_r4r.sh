#!/bin/bash
# visit 4r: why is the 16-wave tap-row-reuse tile (33) five times slower than its 4-wave siblings?  timing-only variants
set -o pipefail
mkdir -p gpurun_out
L=$PWD/yolo-v3-tf2_amd/lib
for v in "" _rs_NO_MASK _rs_NO_EXTRA _rs_NO_SHIFT; do
  Y3_LIB_PATH=$L/liby3hip$v.so timeout -k 10 300 python tools/tune_tiles.py --dtype bf16 --batch 64 --tiles 24,33 --reps 3 > gpurun_out/r4r_sweep$v.txt 2>&1 || { tail -20 gpurun_out/r4r_sweep$v.txt; exit 1; }
  echo "== liby3hip$v"; grep -v amdgpu gpurun_out/r4r_sweep$v.txt | grep -E "k3s1_c(128|256|512)|sum" | awk '{c[$2]++; if (c[$2] <= 1) print}'
done
