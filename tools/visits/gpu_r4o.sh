#!/bin/bash
# visit 4o: bf16 epilogue with the shortcut loads software-pipelined over the row blocks: bf16 tests, then same-box A/B against the
# previous commit's kernel (lib/liby3hip_oldbf16.so)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "bf16" > gpurun_out/4o_tests.log 2>&1 || { tail -40 gpurun_out/4o_tests.log; exit 1; }
tail -2 gpurun_out/4o_tests.log
for rep in 1 2 3; do
  for l in new old; do
    if [ $l = old ]; then export Y3_LIB_PATH=$PWD/yolo-v3-tf2_amd/lib/liby3hip_oldbf16.so; else unset Y3_LIB_PATH; fi
    timeout -k 10 300 python bench.py --dtype bf16 --batch 128 --graph --steps 30 --warmup 10 --no-cpu-baseline --no-sclk > gpurun_out/4o_bf16_${l}_$rep.log 2>&1 || { tail -20 gpurun_out/4o_bf16_${l}_$rep.log; exit 1; }
    echo "$l rep=$rep $(tail -n 1 gpurun_out/4o_bf16_${l}_$rep.log | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
  done
done
unset Y3_LIB_PATH
