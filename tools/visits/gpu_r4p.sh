#!/bin/bash
# visit 4p: upper bound of an LDS-resident halo patch for the bf16 3x3 convs: timing-only build that fetches the activations of the
# first tap only (-DY3_AB_PROBE_A1, wrong results) against the shipped library -- whole conv stack (alternating processes) and per layer
set -o pipefail
mkdir -p gpurun_out
L=yolo-v3-tf2_amd/lib
timeout -k 10 700 python tools/ab_libs.py $L/liby3hip.so $L/liby3hip_probeA1.so --dtype bf16 --batch 128 --rounds 3 > gpurun_out/r4p_ab_bf16_probe_a1.txt 2>&1 || { tail -20 gpurun_out/r4p_ab_bf16_probe_a1.txt; exit 1; }
grep -v amdgpu gpurun_out/r4p_ab_bf16_probe_a1.txt | tail -9
for l in liby3hip liby3hip_probeA1; do
  Y3_LIB_PATH=$PWD/$L/$l.so timeout -k 10 300 python tools/tune_tiles.py --dtype bf16 --batch 64 --tiles 24 --reps 3 > gpurun_out/r4p_sweep_$l.txt 2>&1 || { tail -20 gpurun_out/r4p_sweep_$l.txt; exit 1; }
  echo "== $l"; grep -v amdgpu gpurun_out/r4p_sweep_$l.txt | awk '/k3s1_c128_n256_h52_r1|k3s1_c256_n512_h26_r1|k3s1_c512_n1024_h13_r1|sum/' | awk '{c[$2]++; if (c[$2] <= 2) print}'
done
