#!/bin/bash
# round 5, visit 23: bf16 16x16x32 tiles with the register-direct epilogue (weight rows fetched permuted, no LDS transposition): digests against the LDS-transposed build,
# bf16 GPU tests, alternating A/B of the conv stack, the bf16 bench line
set -o pipefail
mkdir -p gpurun_out
O=gpurun_out/r05_23_bf16_direct_epilogue.txt
: > $O
for lib in liby3hip.so liby3hip_ldsepi.so; do
  echo "== digests $lib (bf16, 128 x 416^2)" >> $O
  Y3_LIB_PATH=$PWD/yolo-v3-tf2_amd/lib/$lib timeout -k 10 300 python tools/hash_outputs.py --dtype bf16 --batch 128 >> $O 2> gpurun_out/r05_23.err || { tail -20 gpurun_out/r05_23.err; exit 1; }
done
python - >> $O <<'PY'
import re
d = {}
cur = None
for ln in open("gpurun_out/r05_23_bf16_direct_epilogue.txt"):
    if ln.startswith("== digests"): cur = ln.split()[2]; d[cur] = []
    elif ln.startswith("DIGEST"): d[cur].append(ln.split()[-1])
a, b = list(d.values())
print("digests equal:", a == b)
PY
tail -1 $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bf16" > gpurun_out/r05_23_tests.log 2>&1 || { tail -60 gpurun_out/r05_23_tests.log; exit 1; }
tail -1 gpurun_out/r05_23_tests.log
timeout -k 10 900 python tools/ab_libs.py yolo-v3-tf2_amd/lib/liby3hip.so yolo-v3-tf2_amd/lib/liby3hip_ldsepi.so --dtype bf16 --batch 128 --rounds 3 >> $O 2> gpurun_out/r05_23.err || { tail -20 gpurun_out/r05_23.err; exit 1; }
tail -8 $O
timeout -k 10 600 python bench.py --dtype bf16 --batch 128 --graph --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r05_23_bench_bf16.json 2> gpurun_out/r05_23_bench.err || { tail -20 gpurun_out/r05_23_bench.err; exit 1; }
python -c "import json; d = json.load(open('gpurun_out/r05_23_bench_bf16.json')); print('bench bf16 graph:', d['value'], 'img/s', d['ms_per_step'], 'ms; lanes', d['config']['lanes'], 'frac', d['roofline']['frac'], 'parity', d['parity_checked'])"
