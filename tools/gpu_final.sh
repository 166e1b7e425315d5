#!/bin/bash
# Round-end measurements on one MI355X: the BASELINE configs through bench.py / bench_configs.py, each to its own log
# under gpurun_out/ (copy the JSON lines into profiles/).  usage: tools/gpu_final.sh <tag>
set -o pipefail
tag=${1:-r02}
mkdir -p gpurun_out
run() { # name, timeout, cmd...
  local name=$1 to=$2; shift 2
  echo "== $name"
  timeout -k 10 "$to" "$@" > "gpurun_out/$name.json" 2> "gpurun_out/$name.err"
  local rc=$?
  echo "== $name rc=$rc"; tail -c 1500 "gpurun_out/$name.json"; echo
  if [ $rc -ge 124 ]; then echo "step $name killed/hung: stopping"; exit $rc; fi
  return 0
}
run ${tag}_bench_f32_b64_s416 500 python bench.py --steps 20 --warmup 5 --per-layer
Y3_BENCH_FORCE_DIST=1 run ${tag}_bench_f32_rccl1 300 python bench.py --steps 10 --warmup 3 --no-alt --no-cpu-baseline
Y3_BENCH_FORCE_DIST=1 run ${tag}_bench_f32_rccl1_graph 300 python bench.py --steps 10 --warmup 3 --no-alt --no-cpu-baseline --graph
run ${tag}_bench_f32_b64_s608 500 python bench.py --steps 10 --warmup 3 --image-size 608 --no-alt --no-cpu-baseline
run ${tag}_bench_bf16_b128_s416 400 python bench.py --steps 20 --warmup 5 --dtype bf16 --batch 128 --graph --no-cpu-baseline
echo "== config2"; timeout -k 10 300 python tools/bench_configs.py > gpurun_out/${tag}_config2_backbone.txt 2>&1; tail -4 gpurun_out/${tag}_config2_backbone.txt
exit 0
