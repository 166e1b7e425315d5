#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
L=yolo-v3-tf2_amd/lib
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "bf16" > gpurun_out/pytest_bf16.log 2>&1 || { tail -40 gpurun_out/pytest_bf16.log; exit 1; }
tail -n 2 gpurun_out/pytest_bf16.log
timeout -k 10 600 python tools/ab_libs.py $L/liby3hip_nopersist.so $L/liby3hip.so $L/liby3hip_resmode1.so --dtype bf16 --batch 128 --rounds 2 > gpurun_out/ab_bf16_persist.log 2>&1 || { tail -20 gpurun_out/ab_bf16_persist.log; exit 1; }
cat gpurun_out/ab_bf16_persist.log
