#!/bin/bash
# round 5, visit 1: (a) sanity of the round's first library (fp32 tile 34 = 32x128 added, bench gate on the fused route, full-size forward_decode test);
# (b) the gate for the residual-block fusion (bf16, 128 x 416^2): what the blocks' 1x1 LAUNCHES cost the step; (c) fp32 probes of VERDICT r04 #3:
# start stagger of lane 1 by 1 / 2 ops, tile 34 on the 13^2 / 26^2 1x1 signatures in the steady-state tuner
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_bench_multirank.py -x -q -m gpu -k "full_size_batch or fused_heads or every_tile_shape or rehearsal or conv_layers" > gpurun_out/r05_01_tests.log 2>&1 || { tail -40 gpurun_out/r05_01_tests.log; exit 1; }
tail -1 gpurun_out/r05_01_tests.log
timeout -k 10 600 python tools/gate_block_fusion.py --dtype bf16 --batch 128 --rounds 3 > gpurun_out/r05_01_gate_bf16.txt 2>&1 || { tail -20 gpurun_out/r05_01_gate_bf16.txt; exit 1; }
grep -v amdgpu gpurun_out/r05_01_gate_bf16.txt | tail -8
L=yolo-v3-tf2_amd/lib/liby3hip.so
timeout -k 10 600 python tools/ab_libs.py "$L" "$L%Y3_LANE_STAGGER=1" "$L%Y3_LANE_STAGGER=2" "$L%Y3_LANE_STAGGER=4" --dtype f32 --rounds 3 > gpurun_out/r05_01_ab_f32_lane_stagger.txt 2>&1 || { tail -20 gpurun_out/r05_01_ab_f32_lane_stagger.txt; exit 1; }
tail -4 gpurun_out/r05_01_ab_f32_lane_stagger.txt
timeout -k 10 900 python tools/tune_steady.py --dtype f32 --batch 64 --only k1s1_c1024_n512_h13,k1s1_c512_n256_h26,k1s1_c512_n256_h13,k1s1_c256_n128_h26,k1s1_c768,k1s1_c256_n128_h52 --tiles 34,32,31,27,26,11,10,5 > gpurun_out/r05_01_tune_f32_tile34.txt 2>&1 || { tail -20 gpurun_out/r05_01_tune_f32_tile34.txt; exit 1; }
tail -9 gpurun_out/r05_01_tune_f32_tile34.txt
