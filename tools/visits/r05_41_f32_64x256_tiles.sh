#!/bin/bash
# round 5, visit 41: fp32 64 x 256 LDS-DMA single-stage tiles (experimental ids 34: wave tile 32 x 128, 4 workgroups per CU; 35: wave tile 64 x 64, 3 per CU): fewer LDS fragment
# reads and L2 -> LDS bytes per MFMA than the 64 x 128 tile at one workgroup per CU less.  Digests, then the conv stack with the tiles on the large 3x3 signatures.
set -o pipefail
mkdir -p gpurun_out
O=gpurun_out/r05_41_f32_64x256_tiles.txt
: > $O
L=yolo-v3-tf2_amd/lib/liby3hip.so
V=yolo-v3-tf2_amd/lib/liby3hip_w128.so
echo "== digests shipped" >> $O
Y3_LIB_PATH=$PWD/$L timeout -k 10 300 python tools/hash_outputs.py --dtype f32 --batch 64 2> gpurun_out/r05_41.err | md5sum >> $O || { tail -20 gpurun_out/r05_41.err; exit 1; }
for t in b c; do
  echo "== digests experimental build, table $t" >> $O
  Y3_LIB_PATH=$PWD/$V Y3_TUNING_FILE=$PWD/tools/tables/f32_b64_s416_w128_$t.json timeout -k 10 300 python tools/hash_outputs.py --dtype f32 --batch 64 2> gpurun_out/r05_41.err | md5sum >> $O || { tail -20 gpurun_out/r05_41.err; exit 1; }
done
cat $O
timeout -k 10 1000 python tools/ab_libs.py $L "$V@tools/tables/f32_b64_s416_w128_a.json" "$V@tools/tables/f32_b64_s416_w128_b.json" "$V@tools/tables/f32_b64_s416_w128_c.json" --dtype f32 --batch 64 --rounds 3 >> $O 2> gpurun_out/r05_41.err || { tail -20 gpurun_out/r05_41.err; cat $O; exit 1; }
tail -14 $O
