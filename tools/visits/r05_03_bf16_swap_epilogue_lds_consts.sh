#!/bin/bash
# round 5, visit 3: the register-direct epilogue again, now with the tile's scale / shift staged in LDS at kernel start (visit 2: the epilogue's
# second-half scale / shift loads sat in the vector-memory queue behind the first half's stores: +4 us per tile, -4 % on the step)
set -o pipefail
mkdir -p gpurun_out
L=yolo-v3-tf2_amd/lib
for lib in liby3hip_base.so liby3hip.so; do
  Y3_LIB_PATH=$PWD/$L/$lib timeout -k 10 300 python tools/hash_outputs.py --dtype bf16 --batch 128 2>/dev/null | grep DIGEST > gpurun_out/r05_03_digest_$lib.txt || { echo "digest run failed for $lib"; exit 1; }
done
if cmp -s gpurun_out/r05_03_digest_liby3hip_base.so.txt gpurun_out/r05_03_digest_liby3hip.so.txt; then echo "DIGESTS EQUAL: bit-identical to the previous build"; else echo "DIGESTS DIFFER"; exit 1; fi
timeout -k 10 700 python tools/ab_libs.py $L/liby3hip_base.so $L/liby3hip.so --dtype bf16 --batch 128 --rounds 3 > gpurun_out/r05_03_ab_bf16_swap.txt 2>&1 || { tail -20 gpurun_out/r05_03_ab_bf16_swap.txt; exit 1; }
grep -v amdgpu gpurun_out/r05_03_ab_bf16_swap.txt | tail -8
