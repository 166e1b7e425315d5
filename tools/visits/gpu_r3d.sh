#!/bin/bash
# round-3 GPU visit D: fused stem kernel -- parity first, then A/B
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fused_stem" > gpurun_out/pytest_stem.log 2>&1 || { tail -40 gpurun_out/pytest_stem.log; exit 1; }
tail -n 3 gpurun_out/pytest_stem.log
timeout -k 10 600 python tools/ab_stem.py > gpurun_out/ab_stem.log 2>&1 || { tail -20 gpurun_out/ab_stem.log; exit 1; }
cat gpurun_out/ab_stem.log
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_all.log 2>&1; tail -n 5 gpurun_out/pytest_all.log
