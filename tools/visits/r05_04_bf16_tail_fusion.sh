#!/bin/bash
# round 5, visit 4: the residual blocks' 1x1 conv 256 -> 128 @52 inside the producing conv's launch (conv_bf16.hip TAIL): bit-identity test against the
# two-launch form, the bf16 suite, then the step with the tail on / off (Y3_TAIL_FUSION) and the round's base build, alternating
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fused_1x1_tail or detect_single_call or forward_decode_fused or nms" > gpurun_out/r05_04_tests_tail.log 2>&1 || { tail -60 gpurun_out/r05_04_tests_tail.log; exit 1; }
tail -1 gpurun_out/r05_04_tests_tail.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bf16" > gpurun_out/r05_04_tests_bf16.log 2>&1 || { tail -60 gpurun_out/r05_04_tests_bf16.log; exit 1; }
tail -1 gpurun_out/r05_04_tests_bf16.log
L=yolo-v3-tf2_amd/lib
timeout -k 10 700 python tools/ab_libs.py $L/liby3hip_base.so "$L/liby3hip.so%Y3_TAIL_FUSION=0" "$L/liby3hip.so%Y3_TAIL_FUSION=1" --dtype bf16 --batch 128 --rounds 3 > gpurun_out/r05_04_ab_bf16_tail.txt 2>&1 || { tail -20 gpurun_out/r05_04_ab_bf16_tail.txt; exit 1; }
grep -v amdgpu gpurun_out/r05_04_ab_bf16_tail.txt | tail -12
