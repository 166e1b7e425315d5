#!/bin/bash
# round 5, visit 36: the fp32 LDS-DMA 64x128 single-stage tile (id 31, 24 KB of LDS) compiled for SIX waves per SIMD (80 VGPRs, 17 spilled to scratch) instead of five (90):
# six workgroups per CU.  Shipped table (tile 31 on the 3x3 128->256 @52 and 64->128 @104 convs) and a table with tile 31 on every large 3x3 signature, both builds, alternating.
set -o pipefail
mkdir -p gpurun_out
O=gpurun_out/r05_36_f32_tile31_six_waves.txt
L=yolo-v3-tf2_amd/lib/liby3hip.so
V=yolo-v3-tf2_amd/lib/liby3hip_t31w6.so
T=tools/tables/f32_b64_s416_more_tile31.json
timeout -k 10 1000 python tools/ab_libs.py $L $V "$L@$T" "$V@$T" --dtype f32 --batch 64 --rounds 3 > $O 2> gpurun_out/r05_36.err || { tail -20 gpurun_out/r05_36.err; cat $O; exit 1; }
cat $O
