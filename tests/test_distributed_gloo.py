"""N>1 path on CPU: world_size-2 gloo.  Each rank runs the per-image pipeline (oracle standing in for the GPU
kernels, tests only) on its shard, packs, all-gathers; the result must equal the single-process result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


_PORT_CLASH = ("EADDRINUSE", "Address already in use", "address already in use", "errno: 98",
               "rendezvous timeout", "Timed out waiting for clients", "timed out after")


def _guarded(target, rank, world, port, q, *args):
    """Worker entry: whatever the body raises goes back to the parent as text (rank, "error", traceback) so that the
    parent can tell a rendezvous port clash (retry) from a crash (fail at once, with the worker's traceback)."""
    try:
        target(rank, world, port, q, *args)
    except BaseException:   # noqa: BLE001 -- reported, then the process exits non-zero
        import traceback
        q.put((rank, "error", traceback.format_exc()))
        raise


def _run_two_ranks(target, n_results, *args):
    """Start two spawned ranks of `target(rank, world, port, q, *args)` and collect n_results queue items.  A second
    attempt is made ONLY when a worker reports a port clash / rendezvous timeout (the probed port can be taken between
    probe and bind); any other worker failure fails the test on the first attempt with that worker's traceback."""
    ctx = mp.get_context("spawn")
    last = ""
    for attempt in range(2):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_guarded, args=(target, r, 2, port, q) + args) for r in range(2)]
        for p in procs:
            p.start()
        items, errors = [], []
        try:
            while len(items) < n_results and not errors:
                it = q.get(timeout=180)
                (errors if (isinstance(it, tuple) and len(it) == 3 and it[1] == "error") else items).append(it)
        except Exception:     # queue.Empty: nobody reported in time
            errors.append((-1, "error", "no result within 180 s (worker hung or died without a report)"))
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
                p.join()
        if not errors and all(p.exitcode == 0 for p in procs):
            return items
        last = "\n".join(f"[rank {e[0]}] {e[2]}" for e in errors) or f"exit codes {[p.exitcode for p in procs]}"
        if not any(tag in last for tag in _PORT_CLASH):
            pytest.fail(f"gloo worker failed (not a port clash, so no retry):\n{last}")
    pytest.fail(f"gloo rendezvous failed twice on fresh ports:\n{last}")


def _pack(bb, cc, ss, sel, nv, M):
    out = np.zeros((len(nv), M, 7), np.int32)
    for b in range(len(nv)):
        n = int(nv[b])
        idx = sel[b, :n]
        out[b, :n, 0:4] = bb[b, idx].view(np.int32)
        out[b, :n, 4] = ss[b, idx].view(np.int32)
        out[b, :n, 5] = cc[b, idx].astype(np.int32)
        out[b, :n, 6] = idx
    return out


def _pipeline(boxes, scores, cls, M):
    from oracle import oracle as O
    sel, nv = O.nms_padded(boxes, scores, M, 0.5, 0.1)
    return _pack(boxes, cls, scores, sel, nv, M), nv


def _worker(rank, world, port, q, n_images):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import yolo_v3_tf2_amd  # noqa: F401
    from tests.helpers import nms_stress_set
    from yolo_v3_tf2_amd.parallel import allgather_detections, allgather_ragged, shard_range
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    boxes, scores = nms_stress_set(np.random.default_rng(5), n_images, 800)
    cls = np.random.default_rng(6).integers(0, 80, scores.shape).astype(np.int64)
    s, e = shard_range(n_images, rank, world)
    packed, nv = _pipeline(boxes[s:e], scores[s:e], cls[s:e], 100)
    tp, tn = torch.from_numpy(packed), torch.from_numpy(nv)
    if n_images % world == 0:
        g, gn = allgather_detections(tp, tn)
    else:
        g, gn = allgather_ragged(tp, tn, n_images)
    if rank == 0:
        q.put((g.numpy(), gn.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_images", [4, 5])
def test_sharded_detect_equals_single_process(n_images):
    from tests.helpers import nms_stress_set
    (g, gn), = _run_two_ranks(_worker, 1, n_images)
    boxes, scores = nms_stress_set(np.random.default_rng(5), n_images, 800)
    cls = np.random.default_rng(6).integers(0, 80, scores.shape).astype(np.int64)
    ref, rn = _pipeline(boxes, scores, cls, 100)
    assert np.array_equal(g, ref) and np.array_equal(gn, rn)
    # unpack view used by consumers of the gathered rows
    from yolo_v3_tf2_amd.runtime import unpack_detections
    pb, ps, pc, pi = unpack_detections(torch.from_numpy(g))
    n0 = int(gn[0])
    assert torch.equal(pi[0, :n0], torch.from_numpy(ref[0, :n0, 6])) and pb.shape == (n_images, 100, 4)


def _comm_worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ["Y3_RCCL_LIB"] = "/nonexistent/librccl.so.0"     # rank 0 cannot draw an id
    import yolo_v3_tf2_amd  # noqa: F401
    from yolo_v3_tf2_amd import _lib
    from yolo_v3_tf2_amd.parallel import Y3Comm
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        Y3Comm.from_torch_distributed()
        out = "no error"
    except _lib.Y3Error as e:
        out = str(e)
    # the ranks are still in step: a collective issued after the failure completes on both
    t = torch.tensor([rank + 1])
    dist.all_reduce(t)
    q.put((rank, out, int(t.item())))
    dist.barrier()
    dist.destroy_process_group()


def test_comm_bootstrap_failure_raises_on_every_rank():
    """Y3Comm.from_torch_distributed with rank 0 unable to draw the RCCL id: every rank raises the same Y3Error and
    nobody is left inside the broadcast (rank 0 used to raise before entering it)."""
    res = sorted(_run_two_ranks(_comm_worker, 2))
    assert [r[0] for r in res] == [0, 1]
    for _, msg, total in res:
        assert "rank 0 could not draw an RCCL unique id" in msg and "librccl not found" in msg, msg
        assert total == 3


def _crashing_worker(rank, world, port, q):
    if rank == 1:
        raise ValueError("deliberate crash of rank 1")
    q.put((rank, "fine"))


def _clashing_worker(rank, world, port, q):
    raise OSError("[Errno 98] EADDRINUSE: Address already in use (simulated)")


def test_retry_tells_a_port_clash_from_a_crash():
    """the retry loop must not hide a worker that crashed once.  A crash fails on the first attempt with
    the worker's traceback; only a rendezvous port clash earns the second attempt."""
    with pytest.raises(pytest.fail.Exception) as ei:
        _run_two_ranks(_crashing_worker, 2)
    assert "not a port clash, so no retry" in str(ei.value) and "deliberate crash of rank 1" in str(ei.value)
    with pytest.raises(pytest.fail.Exception) as ei:
        _run_two_ranks(_clashing_worker, 2)
    assert "failed twice on fresh ports" in str(ei.value)
