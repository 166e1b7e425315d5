#!/bin/bash
# round-3 GPU visit C: which half of E1 costs the 3x3 layers 0.5 %?  + the tail sawtooth of three layers
set -o pipefail
mkdir -p gpurun_out
L=yolo-v3-tf2_amd/lib
timeout -k 10 700 python tools/ab_libs.py $L/liby3hip_base.so $L/liby3hip_pw.so $L/liby3hip_ep.so $L/liby3hip.so --rounds 3 > gpurun_out/ab_e1b.log 2>&1 || { tail -20 gpurun_out/ab_e1b.log; exit 1; }
tail -n 5 gpurun_out/ab_e1b.log
export Y3_LIB_PATH=$PWD/$L/liby3hip_base.so
timeout -k 10 300 python tools/tail_sawtooth.py --cin 128 --cout 256 --size 3 --s 52 --tile 10 --batches 44:80:2 > gpurun_out/saw_52.log 2>&1 || { tail gpurun_out/saw_52.log; exit 1; }
timeout -k 10 300 python tools/tail_sawtooth.py --cin 512 --cout 1024 --size 3 --s 13 --tile 27 --batches 44:80:2 > gpurun_out/saw_13.log 2>&1 || { tail gpurun_out/saw_13.log; exit 1; }
timeout -k 10 300 python tools/tail_sawtooth.py --cin 256 --cout 128 --size 1 --s 52 --tile 17 --batches 44:80:2 > gpurun_out/saw_1x1.log 2>&1 || { tail gpurun_out/saw_1x1.log; exit 1; }
cat gpurun_out/saw_52.log
