#!/bin/bash
# visit 4b: persistent fp32 tiles 33..37 -- correctness (bit-identical to the classic tiles), then the per-layer sweep against the tuned classic tiles
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -s -k "persistent or every_tile_shape or conv0_error_bounded or heuristic_tiles_are" > gpurun_out/r4b_tests.log 2>&1 || { tail -60 gpurun_out/r4b_tests.log; exit 1; }
tail -3 gpurun_out/r4b_tests.log; grep "stem conv1 output" gpurun_out/r4b_tests.log
timeout -k 10 600 python tools/tune_tiles.py --tiles 10,31,26,11,27,32,17,6,33,34,35,36,37 --reps 3 > gpurun_out/r4b_tile_sweep_f32_persistent_b64_s416.txt 2>&1 || { tail -30 gpurun_out/r4b_tile_sweep_f32_persistent_b64_s416.txt; exit 1; }
tail -90 gpurun_out/r4b_tile_sweep_f32_persistent_b64_s416.txt
