#!/bin/bash
# round 5, visit 40: conv-stack rate against the batch on the final library: fp32 with 1 and 2 lanes, bf16 with 3 lanes (the packaged tables of 64 / 128 images for every batch)
set -o pipefail
mkdir -p gpurun_out
O=gpurun_out/r05_40_batch_scaling.txt
: > $O
echo "== fp32, one lane" >> $O
timeout -k 10 500 python tools/batch_scaling.py --dtype f32 --batches 16,32,64,96,128,192 --lanes 1 >> $O 2> gpurun_out/r05_40.err || { tail -20 gpurun_out/r05_40.err; exit 1; }
echo "== fp32, two lanes" >> $O
timeout -k 10 500 python tools/batch_scaling.py --dtype f32 --batches 16,32,64,96,128,192 --lanes 2 >> $O 2> gpurun_out/r05_40.err || { tail -20 gpurun_out/r05_40.err; exit 1; }
echo "== bf16, three lanes (table of 128 images)" >> $O
timeout -k 10 500 python tools/batch_scaling.py --dtype bf16 --table-batch 128 --batches 32,64,128,192,256,384 --lanes 3 --reps 20 >> $O 2> gpurun_out/r05_40.err || { tail -20 gpurun_out/r05_40.err; exit 1; }
cat $O
