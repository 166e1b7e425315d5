#!/bin/bash
# round 5, visit 39: fp32 LDS-DMA tiles with TWO stages of 16 floats (ids 34 / 35, experimental build): the LDS and the barriers per K of the single-stage 32-float tiles 31 / 32,
# but the next K tile requested a tile ahead.  Digests (same k order: must equal the shipped build's), then the conv stack: shipped | 31 -> 34 | every 64x128 / 64x64 signature on 34 / 35
set -o pipefail
mkdir -p gpurun_out
O=gpurun_out/r05_39_f32_kb16_two_stage.txt
: > $O
L=yolo-v3-tf2_amd/lib/liby3hip.so
V=yolo-v3-tf2_amd/lib/liby3hip_kb16.so
TA=tools/tables/f32_b64_s416_kb16_a.json
TB=tools/tables/f32_b64_s416_kb16_b.json
echo "== digests shipped" >> $O
Y3_LIB_PATH=$PWD/$L timeout -k 10 300 python tools/hash_outputs.py --dtype f32 --batch 64 2> gpurun_out/r05_39.err | md5sum >> $O || { tail -20 gpurun_out/r05_39.err; exit 1; }
echo "== digests kb16 build, table b" >> $O
Y3_LIB_PATH=$PWD/$V Y3_TUNING_FILE=$PWD/$TB timeout -k 10 300 python tools/hash_outputs.py --dtype f32 --batch 64 2> gpurun_out/r05_39.err | md5sum >> $O || { tail -20 gpurun_out/r05_39.err; exit 1; }
cat $O
timeout -k 10 1000 python tools/ab_libs.py $L "$V@$TA" "$V@$TB" --dtype f32 --batch 64 --rounds 3 >> $O 2> gpurun_out/r05_39.err || { tail -20 gpurun_out/r05_39.err; cat $O; exit 1; }
tail -12 $O
