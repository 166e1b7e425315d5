#!/bin/bash
# round 5, visit 24: fp32 fused head conv + decode on a 32 x 256 tile of four waves, one LDS stage, four workgroups per CU (was 64 x 256, eight waves, two stages, one
# per CU): digests against the old build, the head / detect GPU tests, alternating A/B of y3_net_forward_decode, the default bench line
set -o pipefail
mkdir -p gpurun_out
O=gpurun_out/r05_24_f32_head_32x256.txt
: > $O
for lib in liby3hip.so liby3hip_head64.so; do
  echo "== digests $lib (f32, 64 x 416^2)" >> $O
  Y3_LIB_PATH=$PWD/yolo-v3-tf2_amd/lib/$lib timeout -k 10 300 python tools/hash_outputs.py --dtype f32 --batch 64 >> $O 2> gpurun_out/r05_24.err || { tail -20 gpurun_out/r05_24.err; exit 1; }
done
python - >> $O <<'PY'
d = {}
cur = None
for ln in open("gpurun_out/r05_24_f32_head_32x256.txt"):
    if ln.startswith("== digests"): cur = ln.split()[2]; d[cur] = []
    elif ln.startswith("DIGEST"): d[cur].append(ln.split()[-1])
a, b = list(d.values())
print("digests equal:", a == b)
PY
tail -1 $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "decode or detect or head or full_size or plugin" > gpurun_out/r05_24_tests.log 2>&1 || { tail -60 gpurun_out/r05_24_tests.log; exit 1; }
tail -1 gpurun_out/r05_24_tests.log
timeout -k 10 900 python tools/ab_libs.py yolo-v3-tf2_amd/lib/liby3hip.so yolo-v3-tf2_amd/lib/liby3hip_head64.so --dtype f32 --batch 64 --route decode --rounds 3 >> $O 2> gpurun_out/r05_24.err || { tail -20 gpurun_out/r05_24.err; exit 1; }
tail -8 $O
timeout -k 10 600 python bench.py --no-alt > gpurun_out/r05_24_bench_f32.json 2> gpurun_out/r05_24_bench.err || { tail -20 gpurun_out/r05_24_bench.err; exit 1; }
python -c "import json; d = json.load(open('gpurun_out/r05_24_bench_f32.json')); print('f32:', d['value'], 'img/s', d['ms_per_step'], 'ms; frac', d['roofline']['frac'], 'clock-limited', d['roofline']['frac_of_clock_limited_peak'], 'sclk', d['roofline']['sclk_mhz'], 'parity', d['parity_checked'])"
