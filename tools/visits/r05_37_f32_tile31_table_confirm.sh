#!/bin/bash
# round 5, visit 37: confirm on another box: the fp32 table with the LDS-DMA tile 31 on every large 3x3 / stride-1 signature at 52^2 / 26^2 (visit 36: +0.18 %, every round) against the shipped table
set -o pipefail
mkdir -p gpurun_out
O=gpurun_out/r05_37_f32_tile31_table_confirm.txt
L=yolo-v3-tf2_amd/lib/liby3hip.so
T=tools/tables/f32_b64_s416_more_tile31.json
timeout -k 10 1000 python tools/ab_libs.py $L "$L@$T" --dtype f32 --batch 64 --rounds 5 --route decode > $O 2> gpurun_out/r05_37.err || { tail -20 gpurun_out/r05_37.err; cat $O; exit 1; }
cat $O
