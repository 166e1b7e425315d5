#!/bin/bash
# visit 4z: rehearsal of bench.py's multi-rank control flow with two ranks on the box's one GPU (gloo, host-copy gather): weak and strong
set -o pipefail
mkdir -p gpurun_out
export Y3_BENCH_REHEARSE_GLOO=1
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 --no-alt > gpurun_out/4z_rehearse_weak.json 2> gpurun_out/4z_rehearse_weak.err || { tail -30 gpurun_out/4z_rehearse_weak.err; exit 1; }
python3 -c 'import json; d=json.loads(open("gpurun_out/4z_rehearse_weak.json").read().strip().splitlines()[-1]); print(d["n_gpus"], d["value"], d["ms_per_step"], d["scaling"], d["config"]["collective"][:60], d["config"]["global_batch"], d["config"]["rank_order_checked"], d["parity_checked"])'
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --global-batch 64 --steps 5 --warmup 2 --no-alt --graph > gpurun_out/4z_rehearse_strong.json 2> gpurun_out/4z_rehearse_strong.err || { tail -30 gpurun_out/4z_rehearse_strong.err; exit 1; }
python3 -c 'import json; d=json.loads(open("gpurun_out/4z_rehearse_strong.json").read().strip().splitlines()[-1]); print(d["n_gpus"], d["value"], d["ms_per_step"], d["scaling"], d["config"]["global_batch"], d["config"]["hip_graph"])'
wc -l gpurun_out/4z_rehearse_weak.json gpurun_out/4z_rehearse_strong.json
