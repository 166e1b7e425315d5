#!/bin/bash
# visit 4c: persistent fp32 tiles with the late table request (counted vmcnt) and the start stagger on / off, against their classic twins
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "persistent" > gpurun_out/r4c_tests.log 2>&1 || { tail -60 gpurun_out/r4c_tests.log; exit 1; }
tail -2 gpurun_out/r4c_tests.log
for st in 0 100 50; do
  Y3_PERS_STAGGER=$st timeout -k 10 600 python tools/tune_tiles.py --tiles 26,31,32,33,34,35,37 --reps 3 > gpurun_out/r4c_sweep_stagger$st.txt 2>&1 || { tail -30 gpurun_out/r4c_sweep_stagger$st.txt; exit 1; }
  echo "== stagger $st"; grep -v amdgpu gpurun_out/r4c_sweep_stagger$st.txt | awk 'NR==1 || /k3s1_c128_n256_h52_r1|k1s1_c256_n128_h52|k3s1_c256_n512_h26_r1|k1s1_c512_n256_h26|k3s1_c512_n1024_h13_r1|k1s1_c1024_n512_h13|k3s1_c64_n128|k3s1_c32|sum/' | awk '{c[$2]++; if (c[$2] <= 2 || NR==1) print}'
done
