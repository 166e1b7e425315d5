#!/bin/bash
# round 5, visit 26: nms_kernel with the batched score scan and the register-resident resolve: the NMS / detect GPU tests, time per call and selection digests against the
# previous build, the bf16 and fp32 bench lines
set -o pipefail
mkdir -p gpurun_out
O=gpurun_out/r05_26_nms_chains.txt
: > $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "nms or detect or pack or plugin or inference or evaluate or full_size" > gpurun_out/r05_26_tests.log 2>&1 || { tail -60 gpurun_out/r05_26_tests.log; exit 1; }
tail -1 gpurun_out/r05_26_tests.log
for lib in liby3hip.so liby3hip_nmsold.so liby3hip.so liby3hip_nmsold.so; do
  for spec in "f32 64" "bf16 128"; do
    set -- $spec
    echo "== $lib $1 $2" >> $O
    Y3_LIB_PATH=$PWD/yolo-v3-tf2_amd/lib/$lib timeout -k 10 300 python tools/time_nms.py --dtype $1 --batch $2 >> $O 2> gpurun_out/r05_26.err || { tail -20 gpurun_out/r05_26.err; exit 1; }
  done
done
cat $O
for lib in liby3hip.so liby3hip_nmsold.so liby3hip.so liby3hip_nmsold.so; do
  Y3_LIB_PATH=$PWD/yolo-v3-tf2_amd/lib/$lib timeout -k 10 600 python bench.py --dtype bf16 --batch 128 --graph --steps 30 --warmup 5 --no-cpu-baseline --parity-images 0 --no-sclk > gpurun_out/r05_26_b.json 2> gpurun_out/r05_26_bench.err || { tail -20 gpurun_out/r05_26_bench.err; exit 1; }
  python -c "import json; d = json.load(open('gpurun_out/r05_26_b.json')); print('$lib bench bf16 graph:', d['value'], 'img/s', d['ms_per_step'], 'ms')" | tee -a $O
done
