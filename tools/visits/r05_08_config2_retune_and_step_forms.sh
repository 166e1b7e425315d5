#!/bin/bash
# round 5, visit 8: (a) bf16 headline under graph replay, the step as ONE C call (NMS + pack per lane) against the composed calls (Y3_BENCH_COMPOSED=1), alternating;
# (b) VERDICT r04 #8: BASELINE config 2 (backbone, 32 x 416^2, fp32): lanes sweep and a steady-state re-tune of the f32_b32 table on the round's kernels
set -o pipefail
mkdir -p gpurun_out
for k in 1 2 3; do
  for form in detect composed; do
    Y3_BENCH_COMPOSED=$([ $form = composed ] && echo 1 || echo 0) timeout -k 10 600 python bench.py --dtype bf16 --batch 128 --graph --steps 30 --warmup 5 --no-cpu-baseline --parity-images 0 --no-sclk > gpurun_out/r05_08_bench_bf16_${form}_$k.json 2> gpurun_out/r05_08_bench.err || { tail -20 gpurun_out/r05_08_bench.err; exit 1; }
    python -c "import json; d = json.load(open('gpurun_out/r05_08_bench_bf16_${form}_$k.json')); print('round $k  $form:', d['value'], 'img/s', d['ms_per_step'], 'ms per step')" | tee -a gpurun_out/r05_08_ab_bf16_step_forms.txt
  done
done
timeout -k 10 300 python tools/bench_configs.py 2>/dev/null | grep "f32 " | tee gpurun_out/r05_08_config2_before.txt
timeout -k 10 600 python tools/lanes_sweep.py --dtype f32 --batch 32 --lanes 1,2,3 2>/dev/null | grep -v amdgpu | tee gpurun_out/r05_08_lanes_sweep_f32_b32.txt
timeout -k 10 1000 python tools/tune_steady.py --dtype f32 --batch 32 --write f32_b32_s416.json > gpurun_out/r05_08_tune_steady_f32_b32.txt 2>&1 || { tail -20 gpurun_out/r05_08_tune_steady_f32_b32.txt; exit 1; }
grep -v amdgpu gpurun_out/r05_08_tune_steady_f32_b32.txt | grep -e "->" -e start -e final
cp yolo-v3-tf2_amd/tuning/f32_b32_s416.json gpurun_out/r05_08_f32_b32_s416.json
timeout -k 10 300 python tools/bench_configs.py 2>/dev/null | grep "f32 " | tee gpurun_out/r05_08_config2_after.txt
