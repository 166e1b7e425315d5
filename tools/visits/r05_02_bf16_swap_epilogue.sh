#!/bin/bash
# round 5, visit 2: the 16x16x32 bf16 tiles with exchanged MFMA operands and a register-direct epilogue (no LDS transposition) against the previous
# build: (a) digests of the whole bf16 network's outputs on the shipped 128-image plan, both builds (bit-identical?), (b) the bf16 parity tests,
# (c) alternating A/B of the conv stack, 128 x 416^2
set -o pipefail
mkdir -p gpurun_out
L=yolo-v3-tf2_amd/lib
for lib in liby3hip_base.so liby3hip.so; do
  Y3_LIB_PATH=$PWD/$L/$lib timeout -k 10 300 python tools/hash_outputs.py --dtype bf16 --batch 128 2>/dev/null | grep DIGEST > gpurun_out/r05_02_digest_$lib.txt || { echo "digest run failed for $lib"; exit 1; }
done
if cmp -s gpurun_out/r05_02_digest_liby3hip_base.so.txt gpurun_out/r05_02_digest_liby3hip.so.txt; then echo "DIGESTS EQUAL: bit-identical to the previous build"; else echo "DIGESTS DIFFER"; diff gpurun_out/r05_02_digest_liby3hip_base.so.txt gpurun_out/r05_02_digest_liby3hip.so.txt | head; fi
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bf16" > gpurun_out/r05_02_tests.log 2>&1 || { tail -40 gpurun_out/r05_02_tests.log; exit 1; }
tail -1 gpurun_out/r05_02_tests.log
timeout -k 10 700 python tools/ab_libs.py $L/liby3hip_base.so $L/liby3hip.so --dtype bf16 --batch 128 --rounds 3 > gpurun_out/r05_02_ab_bf16_swap.txt 2>&1 || { tail -20 gpurun_out/r05_02_ab_bf16_swap.txt; exit 1; }
grep -v amdgpu gpurun_out/r05_02_ab_bf16_swap.txt | tail -8
