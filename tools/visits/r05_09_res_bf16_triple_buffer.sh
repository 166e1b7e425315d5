#!/bin/bash
# round 5, visit 9: conv3x3_res_bf16 (the weight-resident early 3x3 convs of the bf16 path) with THREE patch buffers (the patch of tile i + 2 requested while tile i
# computes; the awaited patch is older than the previous tile's stores) against the two-buffer form of round 4: parity tests, digests, A/B of the conv stack
set -o pipefail
mkdir -p gpurun_out
L=yolo-v3-tf2_amd/lib
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "weight_resident or bf16_every_layer or bf16_full_size or bf16_intermediate" > gpurun_out/r05_09_tests.log 2>&1 || { tail -60 gpurun_out/r05_09_tests.log; exit 1; }
tail -1 gpurun_out/r05_09_tests.log
for lib in liby3hip_base.so liby3hip.so; do
  Y3_LIB_PATH=$PWD/$L/$lib timeout -k 10 300 python tools/hash_outputs.py --dtype bf16 --batch 128 2>/dev/null | grep DIGEST > gpurun_out/r05_09_digest_$lib.txt || { echo "digest run failed for $lib"; exit 1; }
done
if cmp -s gpurun_out/r05_09_digest_liby3hip_base.so.txt gpurun_out/r05_09_digest_liby3hip.so.txt; then echo "DIGESTS EQUAL"; else echo "DIGESTS DIFFER"; exit 1; fi
timeout -k 10 700 python tools/ab_libs.py $L/liby3hip_base.so $L/liby3hip.so --dtype bf16 --batch 128 --rounds 4 > gpurun_out/r05_09_ab_bf16_res3buf.txt 2>&1 || { tail -20 gpurun_out/r05_09_ab_bf16_res3buf.txt; exit 1; }
grep -v amdgpu gpurun_out/r05_09_ab_bf16_res3buf.txt | tail -3
for lib in liby3hip_base.so liby3hip.so; do
  echo "== $lib: per-conv times of one forward (isolated launches, 64 images, one lane), the three weight-resident convs"
  Y3_LIB_PATH=$PWD/$L/$lib timeout -k 10 300 python bench.py --dtype bf16 --batch 64 --steps 3 --warmup 2 --no-cpu-baseline --parity-images 0 --no-sclk --per-layer 2>&1 >/dev/null | grep -E "^conv(3|6|8) "
done | tee gpurun_out/r05_09_per_layer_res.txt
