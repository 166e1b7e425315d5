#!/bin/bash
# round 5, visit 7: (a) the bf16 epilogue with the whole wave tile's shortcut loads requested before its first stores (against -DY3_RES_HOIST=0 = rounds 3-4):
# digests, A/B of the conv stack; (b) per-lane NMS + pack in y3_net_detect: the detect / NMS tests, then the bf16 headline (graph replay, one C call per step)
set -o pipefail
mkdir -p gpurun_out
L=yolo-v3-tf2_amd/lib
for lib in liby3hip_nohoist.so liby3hip.so; do
  Y3_LIB_PATH=$PWD/$L/$lib timeout -k 10 300 python tools/hash_outputs.py --dtype bf16 --batch 128 2>/dev/null | grep DIGEST > gpurun_out/r05_07_digest_$lib.txt || { echo "digest run failed for $lib"; exit 1; }
done
if cmp -s gpurun_out/r05_07_digest_liby3hip_nohoist.so.txt gpurun_out/r05_07_digest_liby3hip.so.txt; then echo "DIGESTS EQUAL"; else echo "DIGESTS DIFFER"; exit 1; fi
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "detect or nms or bf16 or forward_decode" > gpurun_out/r05_07_tests.log 2>&1 || { tail -60 gpurun_out/r05_07_tests.log; exit 1; }
tail -1 gpurun_out/r05_07_tests.log
timeout -k 10 700 python tools/ab_libs.py $L/liby3hip_nohoist.so $L/liby3hip.so --dtype bf16 --batch 128 --rounds 4 > gpurun_out/r05_07_ab_bf16_hoist.txt 2>&1 || { tail -20 gpurun_out/r05_07_ab_bf16_hoist.txt; exit 1; }
grep -v amdgpu gpurun_out/r05_07_ab_bf16_hoist.txt | tail -3
for k in 1 2; do
  timeout -k 10 600 python bench.py --dtype bf16 --batch 128 --graph --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r05_07_bench_bf16_run$k.json 2> gpurun_out/r05_07_bench_bf16_run$k.err || { tail -20 gpurun_out/r05_07_bench_bf16_run$k.err; exit 1; }
  python -c "import json; d = json.load(open('gpurun_out/r05_07_bench_bf16_run$k.json')); print('bench bf16 graph:', d['value'], 'img/s', d['ms_per_step'], 'ms; conv', d['roofline']['ms_per_launch'])"
done
