#!/bin/bash
# visit 5a: the two-rank rehearsal test of bench.py
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_bench_multirank.py -x -q -m gpu > gpurun_out/5a_tests.log 2>&1 || { tail -60 gpurun_out/5a_tests.log; exit 1; }
tail -2 gpurun_out/5a_tests.log
