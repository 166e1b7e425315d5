#!/bin/bash
# round 5, visit 10: bound for a weight-resident fused residual block (1x1 -> LDS -> 3x3 + shortcut from the input patch) at 104^2, where the step is HBM-bound:
# the blocks' 1x1 launches dropped (all but the first) AND the 3x3 convs' shortcut reads dropped; bf16, 128 x 416^2
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python tools/gate_block_fusion.py --dtype bf16 --batch 128 --rounds 4 --free-shortcut > gpurun_out/r05_10_gate_block_104.txt 2>&1 || { tail -20 gpurun_out/r05_10_gate_block_104.txt; exit 1; }
grep -v amdgpu gpurun_out/r05_10_gate_block_104.txt | tail -6
