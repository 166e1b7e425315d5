#!/bin/bash
# visit x: traffic beyond L2 of the whole conv stack with K chunks of 128 channels
set -o pipefail
export TMPDIR=/tmp
for ck in 128; do
  out=gpurun_out/prof_ck$ck
  mkdir -p $out
  export Y3_K_CHUNK=$ck
  BENCH="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt"
  timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE -d $out/pmc3 -o run --output-format csv -- $BENCH > $out/pmc3.log 2>&1 || { tail -5 $out/pmc3.log; exit 1; }
  timeout -k 10 240 rocprofv3 --pmc WRITE_SIZE -d $out/pmc4 -o run --output-format csv -- $BENCH > $out/pmc4.log 2>&1 || { tail -5 $out/pmc4.log; exit 1; }
  python3 tools/traffic_per_layer.py $out > gpurun_out/x_traffic_ck$ck.txt 2>&1 || { tail gpurun_out/x_traffic_ck$ck.txt; exit 1; }
  echo "ck=$ck: $(tail -n 1 gpurun_out/x_traffic_ck$ck.txt)"
done
