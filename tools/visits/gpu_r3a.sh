#!/bin/bash
# round-3 GPU visit A: parity tests -> default bench line (+ per-layer table)
set -o pipefail
mkdir -p gpurun_out
run() { local name=$1 to=$2; shift 2
  echo "== $name" | tee -a gpurun_out/round.log
  timeout -k 10 "$to" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "== $name rc=$rc" | tee -a gpurun_out/round.log
  tail -n ${TAILN:-12} "gpurun_out/$name.log"
  if [ $rc -ge 124 ]; then echo "step $name killed/hung: stopping"; exit $rc; fi
  return $rc
}
: > gpurun_out/round.log
run pytest 900 python -m pytest tests -m gpu -q --timeout 600 -x ${PYTEST_ARGS:-} || exit 1
run bench 600 python bench.py --steps 20 --warmup 5 --per-layer
exit 0
