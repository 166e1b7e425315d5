#!/bin/bash
# round 5, visit 28: can a memory-bound 1x1 conv run ON THE SAME CU as another lane's 256x256 tile?  The 64x64 LDS-DMA tile 11 needs 35 VGPRs and 32 KB of LDS: exactly what
# a 16-wave tile-24 workgroup (4 x 112 of 512 VGPRs per SIMD, 128 of 160 KB) leaves free.  1x1 convs of the residual stages on tile 11, lanes staggered so that one lane's 1x1
# meets the other lanes' 3x3 convs; controls: the same table in lock step, the shipped table staggered.
set -o pipefail
mkdir -p gpurun_out
O=gpurun_out/r05_28_coresident_1x1.txt
L=yolo-v3-tf2_amd/lib/liby3hip.so
T3=tools/tables/bf16_b128_s416_1x1_tile11_lanes3.json
T2=tools/tables/bf16_b128_s416_1x1_tile11_lanes2.json
S2=tools/tables/bf16_b128_s416_shipped_lanes2.json
timeout -k 10 1000 python tools/ab_libs.py $L "$L@$T3" "$L@$T3%Y3_LANE_STAGGER=1" "$L@$T3%Y3_LANE_STAGGER=3" "$L@$T3%Y3_LANE_STAGGER=7" "$L%Y3_LANE_STAGGER=3" "$L@$S2" "$L@$T2" "$L@$T2%Y3_LANE_STAGGER=3" "$L@$T2%Y3_LANE_STAGGER=7" --dtype bf16 --batch 128 --rounds 2 > $O 2> gpurun_out/r05_28.err || { tail -20 gpurun_out/r05_28.err; cat $O; exit 1; }
cat $O
