#!/bin/bash
# round 5, visit 15: probe -- the 256x256 bf16 tile on EIGHT waves (128x64 / 64x128 wave tiles, 16x16x32 MFMAs: 3/4 of the LDS fragment reads per FLOP of the
# 16-wave tile 24): bit-identity, per-layer isolated sweep, steady-state tuner on the 3x3 signatures
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python tools/check_tile_identity.py 24 37 38 2>&1 | grep -v amdgpu | tee gpurun_out/r05_15_identity.txt
timeout -k 10 600 python tools/tune_tiles.py --dtype bf16 --batch 64 --tiles 24,37,38 > gpurun_out/r05_15_tile_sweep_bf16_8wave.txt 2>&1 || { tail -20 gpurun_out/r05_15_tile_sweep_bf16_8wave.txt; exit 1; }
grep -E "k3s" gpurun_out/r05_15_tile_sweep_bf16_8wave.txt | head -20
timeout -k 10 900 python tools/tune_steady.py --dtype bf16 --batch 128 --only k3s1_c128,k3s1_c256,k3s1_c512,k3s2_c128,k3s2_c256,k3s2_c512 --tiles 24,37,38 > gpurun_out/r05_15_tune_steady_bf16_8wave.txt 2>&1 || { tail -20 gpurun_out/r05_15_tune_steady_bf16_8wave.txt; exit 1; }
grep -v amdgpu gpurun_out/r05_15_tune_steady_bf16_8wave.txt | tail -14
