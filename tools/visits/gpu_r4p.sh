#!/bin/bash
# visit 4p: per-phase time stamps inside the 256x256 bf16 tile (diagnostic build), three layer geometries at the lane size
set -o pipefail
mkdir -p gpurun_out
export Y3_LIB_PATH=$PWD/yolo-v3-tf2_amd/lib/liby3hip_stamps.so
: > gpurun_out/4p_phase_stamps.txt
for geo in "128 256 52" "256 512 26" "512 1024 13"; do
  set -- $geo
  echo "## 3x3 $1 -> $2 @$3, 64 images, with shortcut" >> gpurun_out/4p_phase_stamps.txt
  timeout -k 10 200 python tools/phase_stamps.py --cin $1 --cout $2 --s $3 --batch 64 >> gpurun_out/4p_phase_stamps.txt 2>&1 || { tail -20 gpurun_out/4p_phase_stamps.txt; exit 1; }
done
echo "## 3x3 128 -> 256 @52, 64 images, no shortcut" >> gpurun_out/4p_phase_stamps.txt
timeout -k 10 200 python tools/phase_stamps.py --cin 128 --cout 256 --s 52 --batch 64 --shortcut 0 >> gpurun_out/4p_phase_stamps.txt 2>&1 || { tail -20 gpurun_out/4p_phase_stamps.txt; exit 1; }
cat gpurun_out/4p_phase_stamps.txt
