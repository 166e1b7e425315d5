#!/bin/bash
# visit 4q: whole GPU suite on HEAD, then the round-end measurements (tools/gpu_final.sh r03)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/4q_tests.log 2>&1 || { tail -30 gpurun_out/4q_tests.log; exit 1; }
tail -2 gpurun_out/4q_tests.log
bash tools/gpu_final.sh r03 > gpurun_out/4q_final.log 2>&1 || { tail -30 gpurun_out/4q_final.log; exit 1; }
grep "rc=" gpurun_out/4q_final.log
for f in r03_bench_f32_b64_s416 r03_bench_f32_b64_s608 r03_bench_bf16_b128_s416 r03_bench_f32_rccl1 r03_bench_f32_rccl1_graph; do
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/$f.json').read().strip().splitlines()[-1])
print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('sclk_mhz'), d.get('parity',{}).get('end_to_end_selection_equal'))"
done
tail -4 gpurun_out/r03_config2_backbone.txt
