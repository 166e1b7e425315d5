#!/bin/bash
# visit 4m: fp32 weight-resident conv3 (tile 33): bit-identity tests, then the conv stack with it against the generic 64x64 tile on conv3 (one box, alternating)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -k "weight_resident or conv_layers or network_grids or full_size_batch_properties or fused_stem_matches or lanes_bit or forward_is_det" > gpurun_out/r4m_tests.log 2>&1 || { tail -60 gpurun_out/r4m_tests.log; exit 1; }
tail -2 gpurun_out/r4m_tests.log
for t in generic resident generic resident generic resident; do
  if [ $t = resident ]; then unset Y3_TUNING_FILE; else export Y3_TUNING_FILE=$PWD/tools/tables/f32_b64_s416_generic_conv3.json; fi
  timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-alt --no-cpu-baseline --parity-images 0 --no-sclk > gpurun_out/r4m_f32_$t.json 2> gpurun_out/r4m_f32_$t.err || { tail -20 gpurun_out/r4m_f32_$t.err; exit 1; }
  python - <<PY
import json
d = json.load(open("gpurun_out/r4m_f32_$t.json"))
print("$t", d["value"], d["ms_per_step"], d["roofline"]["ms_per_launch"], d["roofline"]["frac"])
PY
done
timeout -k 10 300 python tools/tune_tiles.py --tiles 11,32,33 --reps 3 > gpurun_out/r4m_sweep_f32_resident.txt 2>&1 || { tail -30 gpurun_out/r4m_sweep_f32_resident.txt; exit 1; }
grep -v amdgpu gpurun_out/r4m_sweep_f32_resident.txt | awk 'NR==1 || /k3s1_c32/'
