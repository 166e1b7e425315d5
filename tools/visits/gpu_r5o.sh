#!/bin/bash
# visit 5o: how many fp32 workgroups does a CU hold at once? (phase stamps + HW_ID, tiles 31 / 10 / 27 / 17)
set -o pipefail
mkdir -p gpurun_out
export Y3_LIB_PATH=$PWD/yolo-v3-tf2_amd/lib/liby3hip_stamps.so
: > gpurun_out/5o_residency.txt
run() { echo "## $1" >> gpurun_out/5o_residency.txt; shift; timeout -k 10 200 python tools/phase_stamps.py --dtype f32 --batch 64 "$@" 2>&1 | grep -v amdgpu.ids >> gpurun_out/5o_residency.txt || { tail -20 gpurun_out/5o_residency.txt; exit 1; }; }
run "3x3 128 -> 256 @52 + shortcut, tile 31 (64x128, 1 stage, LDS-DMA)" --cin 128 --cout 256 --s 52 --tile 31
run "3x3 128 -> 256 @52 + shortcut, tile 10 (64x128, 1 stage)" --cin 128 --cout 256 --s 52 --tile 10
run "3x3 512 -> 1024 @13 + shortcut, tile 27 (64x64, 2 stages, LDS-DMA)" --cin 512 --cout 1024 --s 13 --tile 27
run "1x1 256 -> 128 @52, tile 17 (128x64, 8 waves)" --cin 128 --cout 256 --s 52 --size 1 --tile 17
grep -E "^##|resident|whole workgroup|entry ->" gpurun_out/5o_residency.txt
