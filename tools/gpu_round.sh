#!/bin/bash
# One GPU-box visit: smoke -> parity tests -> short bench.  Stops at the first step that is killed or hangs.
set -o pipefail
mkdir -p gpurun_out
run() { # name, timeout, cmd...
  local name=$1 to=$2; shift 2
  echo "== $name" | tee -a gpurun_out/round.log
  timeout -k 10 "$to" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "== $name rc=$rc" | tee -a gpurun_out/round.log
  tail -n 15 "gpurun_out/$name.log"
  if [ $rc -ge 124 ]; then echo "step $name killed/hung: stopping"; exit $rc; fi
  return $rc
}
: > gpurun_out/round.log
run smoke 300 python __graft_entry__.py smoke
run pytest 900 python -m pytest tests -m gpu -q -x --timeout 600 ${PYTEST_ARGS:-}
run bench 600 python bench.py --steps ${STEPS:-5} --warmup 2 --per-layer ${BENCH_ARGS:-}
exit 0
