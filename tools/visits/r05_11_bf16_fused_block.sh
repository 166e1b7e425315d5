#!/bin/bash
# round 5, visit 11: the early residual blocks of the bf16 path as one launch each (conv_block_bf16.hip): bit-identity test against the two-launch form, the bf16
# suite, then the step with the block fusion on / off (Y3_BLOCK_FUSION), alternating
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_block" > gpurun_out/r05_11_tests_block.log 2>&1 || { tail -60 gpurun_out/r05_11_tests_block.log; exit 1; }
tail -1 gpurun_out/r05_11_tests_block.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bf16" > gpurun_out/r05_11_tests_bf16.log 2>&1 || { tail -60 gpurun_out/r05_11_tests_bf16.log; exit 1; }
tail -1 gpurun_out/r05_11_tests_bf16.log
L=yolo-v3-tf2_amd/lib
for m in 0 1; do
  Y3_BLOCK_FUSION=$m timeout -k 10 300 python tools/hash_outputs.py --dtype bf16 --batch 128 2>/dev/null | grep DIGEST > gpurun_out/r05_11_digest_block$m.txt || { echo "digest run failed"; exit 1; }
done
if cmp -s gpurun_out/r05_11_digest_block0.txt gpurun_out/r05_11_digest_block1.txt; then echo "DIGESTS EQUAL: the whole network, fused blocks vs two launches"; else echo "DIGESTS DIFFER"; exit 1; fi
timeout -k 10 700 python tools/ab_libs.py "$L/liby3hip.so%Y3_BLOCK_FUSION=0" "$L/liby3hip.so%Y3_BLOCK_FUSION=1" --dtype bf16 --batch 128 --rounds 4 > gpurun_out/r05_11_ab_bf16_block.txt 2>&1 || { tail -20 gpurun_out/r05_11_ab_bf16_block.txt; exit 1; }
grep -v amdgpu gpurun_out/r05_11_ab_bf16_block.txt | tail -3
