"""bench.py with more than one rank: the control flow the driver's 8-GPU run goes through (sharding, the rank-order gate, barriers,
max-over-ranks timing, one JSON line from rank 0), rehearsed with two ranks that SHARE the box's one GPU.  RCCL needs one GPU per rank,
so the rehearsal replaces the exchange by a gloo all-gather of host copies (Y3_BENCH_REHEARSE_GLOO=1) -- the line says so and is not a
measurement.  The RCCL route itself is covered with one rank by test_gpu_parity.py::test_y3_comm_allgather_single_rank_and_graph."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
@pytest.mark.parametrize("extra,scaling,global_batch", [([], "weak", 8), (["--global-batch", "6"], "strong", 6)])
def test_bench_two_ranks_rehearsal(extra, scaling, global_batch):
    env = dict(os.environ, Y3_BENCH_REHEARSE_GLOO="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "4", "--image-size", "96",
           "--steps", "3", "--warmup", "1", "--no-alt", "--no-cpu-baseline", "--no-sclk"] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]          # exactly one JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == scaling and d["config"]["global_batch"] == global_batch
    # the rehearsal exchanges over gloo: no rank touched RCCL, and the line says so in the one field a reader checks for it
    assert d["config"]["rank_order_checked"] is True and d["config"]["rccl_ranks"] == 0 and d["config"]["rehearsal"] is True
    assert d["config"]["collective"].startswith("REHEARSAL")
    assert d["parity"]["timed_route_equals_composed_route"] is True
    assert d["parity_checked"] == 2 and d["parity"]["nms_index_selection"].startswith("bit-exact")
    assert d["value"] > 0 and abs(d["value"] - global_batch * d["steps"] / (d["ms_per_step"] * d["steps"] / 1e3)) <= 0.02 * d["value"]


def _bench(args, env_extra=None, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, env=env, capture_output=True,
                          text=True, timeout=timeout)


def test_gpus_n_without_enough_gpus_exits_nonzero_naming_the_shortfall():
    """`python bench.py --gpus 8` used to run ONE rank and print n_gpus 1 with rc 0.  Now --gpus N means
    N ranks on N GPUs or no number.  Runs anywhere: more ranks than any box has GPUs."""
    r = _bench(["--gpus", "64", "--steps", "1", "--warmup", "1"])
    assert r.returncode != 0
    assert "--gpus 64 needs 64 GPUs" in r.stderr and "no number reported" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]


def test_gpus_n_must_equal_the_launchers_world_size():
    r = _bench(["--gpus", "8", "--steps", "1", "--warmup", "1"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "--gpus 8 but the launcher started WORLD_SIZE=2" in r.stderr
    assert not r.stdout.strip()
    r = _bench(["--gpus", "1", "--steps", "1", "--warmup", "1"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "--gpus 1 but the launcher started WORLD_SIZE=2" in r.stderr


@pytest.mark.gpu
def test_bench_launcherless_two_ranks_rehearsal():
    """`python bench.py --gpus 2` with no launcher: bench.py starts the two ranks itself (child process), relays ONE JSON line."""
    r = _bench(["--gpus", "2", "--batch", "4", "--image-size", "96", "--steps", "3", "--warmup", "1", "--no-alt",
                "--no-cpu-baseline", "--no-sclk"], {"Y3_BENCH_REHEARSE_GLOO": "1"})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["rccl_ranks"] == 0 and d["config"]["rehearsal"] is True and d["config"]["rank_order_checked"] is True
    assert d["config"]["collective"].startswith("REHEARSAL") and d["parity_checked"] == 2


@pytest.mark.gpu
def test_bench_gpus_2_on_a_one_gpu_box_fails_loudly():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with one GPU")
    r = _bench(["--gpus", "2", "--steps", "1", "--warmup", "1"])
    assert r.returncode != 0 and "--gpus 2 needs 2 GPUs" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
