#!/bin/bash
# visit 4aa: timing-only: bf16 generic kernel, 3x3 convs, operand fetches of K tiles 2.. removed (the MFMAs and fragment reads run on stale
# LDS contents; wrong results): what the K loop does when no fetch latency / L2 bandwidth is in its way
set -o pipefail
mkdir -p gpurun_out
L=yolo-v3-tf2_amd/lib
timeout -k 10 700 python tools/ab_libs.py $L/liby3hip.so $L/liby3hip_nofetch.so --dtype bf16 --batch 128 --rounds 2 > gpurun_out/r4aa_ab_bf16_nofetch.txt 2>&1 || { tail -20 gpurun_out/r4aa_ab_bf16_nofetch.txt; exit 1; }
grep -v amdgpu gpurun_out/r4aa_ab_bf16_nofetch.txt | tail -2
for l in liby3hip liby3hip_nofetch; do
  Y3_LIB_PATH=$PWD/$L/$l.so timeout -k 10 300 python tools/tune_tiles.py --dtype bf16 --batch 64 --tiles 24 --reps 3 > gpurun_out/r4aa_sweep_$l.txt 2>&1 || { tail -20 gpurun_out/r4aa_sweep_$l.txt; exit 1; }
  echo "== $l"; grep -v amdgpu gpurun_out/r4aa_sweep_$l.txt | grep -E "k3s[12]_c(128|256|512)|sum" | awk '{c[$2]++; if (c[$2] <= 1) print}'
done
