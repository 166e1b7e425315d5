#!/bin/bash
# GPU visit: parity tests -> fp32 tile sweep incl. the stream-K tiles -> bench.  Stops at the first step that is killed.
set -o pipefail
mkdir -p gpurun_out
run() { # name, timeout, cmd...
  local name=$1 to=$2; shift 2
  echo "== $name" | tee -a gpurun_out/round.log
  timeout -k 10 "$to" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "== $name rc=$rc" | tee -a gpurun_out/round.log
  tail -n ${TAILN:-12} "gpurun_out/$name.log"
  if [ $rc -ge 124 ]; then echo "step $name killed/hung: stopping"; exit $rc; fi
  return $rc
}
: > gpurun_out/round.log
run pytest 900 python -m pytest tests -m gpu -q --timeout 600 ${PYTEST_ARGS:-}
run sweep_sk 600 python tools/tune_tiles.py --batch 64 --image-size 416 --reps 3 --tiles 10,11,9,17,26,27,31,32,33,34,35,36,37,38,39,40
run bench 600 python bench.py --steps ${STEPS:-10} --warmup 3 --per-layer --no-alt ${BENCH_ARGS:-}
exit 0
