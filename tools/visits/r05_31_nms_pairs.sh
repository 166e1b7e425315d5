#!/bin/bash
# round 5, visit 31: nms_kernel with the kept-list test spread over all threads and the balanced row build (against the build of visit 26): the NMS / detect GPU tests, time per call and selection digests against the
# previous build, the bf16 and fp32 bench lines
set -o pipefail
mkdir -p gpurun_out
O=gpurun_out/r05_31_nms_chains.txt
: > $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "nms or detect or pack or plugin or inference or evaluate or full_size" > gpurun_out/r05_31_tests.log 2>&1 || { tail -60 gpurun_out/r05_31_tests.log; exit 1; }
tail -1 gpurun_out/r05_31_tests.log
for lib in liby3hip.so liby3hip_nms1.so liby3hip.so liby3hip_nms1.so; do
  for spec in "f32 64" "bf16 128"; do
    set -- $spec
    echo "== $lib $1 $2" >> $O
    Y3_LIB_PATH=$PWD/yolo-v3-tf2_amd/lib/$lib timeout -k 10 300 python tools/time_nms.py --dtype $1 --batch $2 >> $O 2> gpurun_out/r05_31.err || { tail -20 gpurun_out/r05_31.err; exit 1; }
  done
done
cat $O
for lib in liby3hip.so liby3hip_nms1.so liby3hip.so liby3hip_nms1.so; do
  Y3_LIB_PATH=$PWD/yolo-v3-tf2_amd/lib/$lib timeout -k 10 600 python bench.py --dtype bf16 --batch 128 --graph --steps 30 --warmup 5 --no-cpu-baseline --parity-images 0 --no-sclk > gpurun_out/r05_31_b.json 2> gpurun_out/r05_31_bench.err || { tail -20 gpurun_out/r05_31_bench.err; exit 1; }
  python -c "import json; d = json.load(open('gpurun_out/r05_31_b.json')); print('$lib bench bf16 graph:', d['value'], 'img/s', d['ms_per_step'], 'ms')" | tee -a $O
done
