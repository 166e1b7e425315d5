#!/bin/bash
# round 5, visit 6: where a 256x256 tile with the fused 1x1 tail spends its time (diagnostic build with per-workgroup clock stamps), against the same conv
# without the tail and the 1x1's own launch; 64 images (one lane), 128 -> 256 @52 + shortcut
set -o pipefail
mkdir -p gpurun_out
export Y3_LIB_PATH=$PWD/yolo-v3-tf2_amd/lib/liby3hip_stamps.so
{ echo "== 3x3 128 -> 256 @52 + shortcut, the 1x1 256 -> 128 behind it FUSED into its launch"; Y3_TAIL_FUSION=1 timeout -k 10 300 python tools/phase_stamps.py --tail 1 2>/dev/null | grep -v amdgpu;
  echo "== the same 3x3, the 1x1 as its own launch (tail fusion off)"; Y3_TAIL_FUSION=0 timeout -k 10 300 python tools/phase_stamps.py --tail 1 2>/dev/null | grep -v amdgpu;
  echo "== 26^2: 256 -> 512 + shortcut (no tail possible)"; timeout -k 10 300 python tools/phase_stamps.py --cin 256 --cout 512 --s 26 2>/dev/null | grep -v amdgpu; } > gpurun_out/r05_06_tail_phase_stamps.txt 2>&1 || { tail -30 gpurun_out/r05_06_tail_phase_stamps.txt; exit 1; }
cat gpurun_out/r05_06_tail_phase_stamps.txt
