#!/bin/bash
# round 5, visit 29: conv_f32.hip compiled with LLVM's alternative AMDGPU scheduling strategies (-mllvm -amdgpu-sched-strategy=...): same source, other instruction order in
# prologue / K loop / epilogue; digests against the default build, then alternating A/B of the fp32 conv stack
set -o pipefail
mkdir -p gpurun_out
O=gpurun_out/r05_29_f32_sched_strategies.txt
: > $O
D=yolo-v3-tf2_amd/lib
for lib in liby3hip.so liby3hip_f32_max-ilp.so liby3hip_f32_max-memory-clause.so liby3hip_f32_iterative-ilp.so liby3hip_f32_iterative-minreg.so; do
  echo "== digests $lib" >> $O
  Y3_LIB_PATH=$PWD/$D/$lib timeout -k 10 300 python tools/hash_outputs.py --dtype f32 --batch 64 2> gpurun_out/r05_29.err | md5sum >> $O || { tail -20 gpurun_out/r05_29.err; exit 1; }
done
timeout -k 10 1000 python tools/ab_libs.py $D/liby3hip.so $D/liby3hip_f32_max-ilp.so $D/liby3hip_f32_max-memory-clause.so $D/liby3hip_f32_iterative-ilp.so $D/liby3hip_f32_iterative-minreg.so --dtype f32 --batch 64 --rounds 3 >> $O 2> gpurun_out/r05_29.err || { tail -20 gpurun_out/r05_29.err; cat $O; exit 1; }
cat $O
