#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
L=yolo-v3-tf2_amd/lib
timeout -k 10 600 python tools/ab_libs.py $L/liby3hip.so $L/liby3hip_nostore.so $L/liby3hip_nostore_nores.so $L/liby3hip_noepi.so --dtype bf16 --batch 128 --rounds 2 > gpurun_out/ab_bf16_noepi.log 2>&1 || { tail -20 gpurun_out/ab_bf16_noepi.log; exit 1; }
cat gpurun_out/ab_bf16_noepi.log
