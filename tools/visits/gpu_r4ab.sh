#!/bin/bash
# visit 4ab: feasibility: the generic bf16 kernel instantiated with 4 waves of 128x128 (one wave per SIMD, 256 accumulator registers; LDS reads
# per MFMA halved against the 16-wave tile), compiler-scheduled: tiles 36 (32x32x16) / 37 (16x16x32) against 14 (8 waves, 128x64) and 24
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python tools/tune_tiles.py --dtype bf16 --batch 64 --tiles 14,24,36,37 --reps 3 > gpurun_out/r4ab_sweep_b64.txt 2>&1 || { tail -20 gpurun_out/r4ab_sweep_b64.txt; exit 1; }
grep -v amdgpu gpurun_out/r4ab_sweep_b64.txt | grep -E "k3s[12]_c(128|256|512)|k1s1_c1024_n512|conv  shape|sum" | awk '{c[$2]++; if (c[$2] <= 1 || $1 == "conv") print}'
