#!/bin/bash
# visit 4v: the whole GPU suite + smoke + the default bench line on the library as committed (tap-row reuse tiles in, probe ifdef out)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r4v_tests.txt 2>&1 || { tail -40 gpurun_out/r4v_tests.txt; exit 1; }
tail -3 gpurun_out/r4v_tests.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4v_smoke.txt 2>&1 || { tail -20 gpurun_out/r4v_smoke.txt; exit 1; }
tail -2 gpurun_out/r4v_smoke.txt
timeout -k 10 500 python bench.py > gpurun_out/r4v_bench_default.json 2> gpurun_out/r4v_bench_default.err || { tail -20 gpurun_out/r4v_bench_default.err; exit 1; }
tail -c 1200 gpurun_out/r4v_bench_default.json
