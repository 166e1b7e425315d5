#!/bin/bash
# round 5, visit 30: conv_bf16.hip compiled with LLVM's alternative AMDGPU scheduling strategies: digests against the default build, alternating A/B of the bf16 conv stack
set -o pipefail
mkdir -p gpurun_out
O=gpurun_out/r05_30_bf16_sched_strategies.txt
: > $O
D=yolo-v3-tf2_amd/lib
for lib in liby3hip.so liby3hip_bf16_max-ilp.so liby3hip_bf16_max-memory-clause.so liby3hip_bf16_iterative-ilp.so liby3hip_bf16_iterative-minreg.so; do
  echo "== digests $lib" >> $O
  Y3_LIB_PATH=$PWD/$D/$lib timeout -k 10 300 python tools/hash_outputs.py --dtype bf16 --batch 128 2> gpurun_out/r05_30.err | md5sum >> $O || { tail -20 gpurun_out/r05_30.err; exit 1; }
done
timeout -k 10 1000 python tools/ab_libs.py $D/liby3hip.so $D/liby3hip_bf16_max-ilp.so $D/liby3hip_bf16_max-memory-clause.so $D/liby3hip_bf16_iterative-ilp.so $D/liby3hip_bf16_iterative-minreg.so --dtype bf16 --batch 128 --rounds 3 >> $O 2> gpurun_out/r05_30.err || { tail -20 gpurun_out/r05_30.err; cat $O; exit 1; }
cat $O
