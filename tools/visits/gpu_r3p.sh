#!/bin/bash
# round-3 GPU visit P: the multi-rank code paths of bench.py with one rank (RCCL forced on), torchrun launch, smoke()
set -o pipefail
mkdir -p gpurun_out
run() { local name=$1; shift; timeout -k 10 300 "$@" > gpurun_out/$name.log 2>&1 || { echo "$name FAILED"; tail -25 gpurun_out/$name.log; exit 1; }; tail -n 1 gpurun_out/$name.log | cut -c1-700; }
Y3_BENCH_FORCE_DIST=1 run dist1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt
Y3_BENCH_FORCE_DIST=1 run dist1_graph python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt --graph
Y3_BENCH_FORCE_DIST=1 run dist1_torch python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt --collective torch
Y3_BENCH_FORCE_DIST=1 run dist1_global python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt --global-batch 32
run torchrun1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-alt
run smoke python __graft_entry__.py smoke
