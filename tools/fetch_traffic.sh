#!/bin/bash
# FETCH_SIZE / WRITE_SIZE / L2 hit passes only (HBM-side traffic per kernel), summarised like tools/profile.sh.
# usage: tools/fetch_traffic.sh <tag> [bench args...]  -> gpurun_out/prof_<tag>/summary_{derived.txt,traffic.json}
tag=${1:-x}; shift
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt --parity-images 0 $*"
timeout -k 10 240 rocprofv3 --pmc FETCH_SIZE -d $out/pmc3 -o run --output-format csv -- $BENCH > $out/pmc3.log 2>&1 || tail -5 $out/pmc3.log
timeout -k 10 240 rocprofv3 --pmc WRITE_SIZE -d $out/pmc4 -o run --output-format csv -- $BENCH > $out/pmc4.log 2>&1 || tail -5 $out/pmc4.log
timeout -k 10 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $out/pmc5 -o run --output-format csv -- $BENCH > $out/pmc5.log 2>&1 || tail -5 $out/pmc5.log
python3 tools/summarize_prof.py $out
cat $out/summary_derived.txt | head -16; cat $out/summary_traffic.json
