#!/bin/bash
# visit 4e: three prologues of the classic fp32 kernel on one box: arithmetic (r03), arithmetic + lean 1x1 (compile-time), row tables + lean 1x1
set -o pipefail
mkdir -p gpurun_out
L=yolo-v3-tf2_amd/lib
timeout -k 10 900 python tools/ab_libs.py $L/liby3hip_arith.so $L/liby3hip_arith_lean.so $L/liby3hip.so --rounds 4 > gpurun_out/r4e_ab_prologues.txt 2>&1 || { tail -20 gpurun_out/r4e_ab_prologues.txt; exit 1; }
grep -v amdgpu gpurun_out/r4e_ab_prologues.txt | tail -16
