#!/bin/bash
# visit 5n: the GPU suite against the experimental build (stream-K, residual prefetch, probes, pipelined bf16 tile compiled in)
set -o pipefail
mkdir -p gpurun_out
export Y3_LIB_PATH=$PWD/yolo-v3-tf2_amd/lib/liby3hip_exp.so
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/5n_tests_exp.log 2>&1 || { tail -30 gpurun_out/5n_tests_exp.log; exit 1; }
tail -2 gpurun_out/5n_tests_exp.log
