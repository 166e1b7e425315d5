"""Multi-GPU data parallelism for the detect path: one process per GPU, images sharded by rank, weights
replicated, and ONE exchange per batch -- an all-gather of the packed detections over RCCL/xGMI.

The reference has no multi-device code at all; images are independent end to end
(conv, decode and NMS are per image), so there is no data-path collective other than gathering the final
[B,100,7] packed rows (+ num_valid).  `torch.distributed` backend "nccl" is RCCL on ROCm; the same code
runs on CPU tensors with the "gloo" backend (used by the CPU test tier).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_images: int, rank: int, world: int) -> Tuple[int, int]:
    """[start, stop) of the images owned by `rank`; the first n % world ranks own one extra image."""
    if world <= 0 or not (0 <= rank < world) or n_images < 0:
        raise ValueError("bad shard arguments")
    q, r = divmod(n_images, world)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def allgather_detections(packed: torch.Tensor, num_valid: torch.Tensor, group=None,
                         out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
    """packed [b,M,7] int32, num_valid [b] int32 (equal b on every rank) -> ([world*b,M,7], [world*b]) in
    rank order.  Single-process (no process group): returns the inputs."""
    if not (dist.is_available() and dist.is_initialized()):
        return packed, num_valid
    world = dist.get_world_size(group)
    if out is None:
        out = (torch.empty((world * packed.shape[0],) + tuple(packed.shape[1:]), dtype=packed.dtype,
                           device=packed.device),
               torch.empty((world * num_valid.shape[0],), dtype=num_valid.dtype, device=num_valid.device))
    dist.all_gather_into_tensor(out[0], packed.contiguous(), group=group)
    dist.all_gather_into_tensor(out[1], num_valid.contiguous(), group=group)
    return out


class Y3Comm:
    """RCCL communicator behind the C ABI (y3_comm_* / y3_allgather_results in include/y3.h): the same exchange as
    `allgather_detections`, but enqueued by liby3hip.so itself on the caller's stream -- one RCCL group per batch that
    can sit in the same HIP graph as y3_net_detect, and that a non-Python host can call.

    `Y3Comm.from_torch_distributed()` bootstraps from an initialised torch.distributed process group: rank 0 draws the
    unique id and broadcasts it (128 bytes); the communicator itself is RCCL's own, not torch's.  One rank per GPU,
    the current device is the rank's GPU."""

    def __init__(self, unique_id: bytes, world: int, rank: int):
        import ctypes as C
        from . import _lib
        _lib.require_gpu()
        if len(unique_id) != _lib.Y3_COMM_ID_BYTES:
            raise ValueError("unique id must be 128 bytes")
        self.lib = _lib.load()
        self.world, self.rank = int(world), int(rank)
        self._h = C.c_void_p()
        buf = C.create_string_buffer(bytes(unique_id), _lib.Y3_COMM_ID_BYTES)
        _lib.check(self.lib.y3_comm_init_rank(buf, self.world, self.rank, C.byref(self._h)), "y3_comm_init_rank")

    @staticmethod
    def new_unique_id() -> bytes:
        import ctypes as C
        from . import _lib
        buf = C.create_string_buffer(_lib.Y3_COMM_ID_BYTES)
        _lib.check(_lib.load().y3_comm_get_unique_id(buf), "y3_comm_get_unique_id")
        return buf.raw

    @classmethod
    def from_torch_distributed(cls, group=None):
        """Every rank enters the same two collectives whatever happens on any of them: rank 0 broadcasts either the id
        or the error it hit drawing it, and after ncclCommInitRank the ranks agree (all-reduce of a flag) that every
        one of them holds a communicator -- so a failure raises the same exception everywhere instead of leaving some
        ranks inside a collective the others never join."""
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        box = [None]
        if rank == 0:
            try:
                box[0] = ("id", cls.new_unique_id())
            except Exception as e:   # noqa: BLE001 - forwarded to every rank below
                box[0] = ("err", f"{type(e).__name__}: {e}")
        dist.broadcast_object_list(box, src=0, group=group)
        kind, payload = box[0]
        if kind != "id":
            from . import _lib
            raise _lib.Y3Error(f"rank 0 could not draw an RCCL unique id: {payload}")
        comm, err = None, ""
        try:
            comm = cls(payload, world, rank)
        except Exception as e:   # noqa: BLE001
            err = f"{type(e).__name__}: {e}"
        backend = dist.get_backend(group)
        dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
        bad = torch.tensor([0 if comm is not None else 1], dtype=torch.int32, device=dev)
        dist.all_reduce(bad, group=group)
        if int(bad.item()):
            if comm is not None:
                comm.close()
            from . import _lib
            raise _lib.Y3Error(f"y3_comm_init_rank failed on {int(bad.item())} of {world} ranks"
                               + (f" (this rank: {err})" if err else ""))
        return comm

    def allgather(self, packed: torch.Tensor, num_valid: torch.Tensor,
                  out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
        """packed [b,M,7] int32, num_valid [b] int32 on the GPU -> ([world*b,M,7], [world*b]) in rank order."""
        import ctypes as C
        from . import _lib
        if not (packed.is_cuda and num_valid.is_cuda and packed.is_contiguous() and num_valid.is_contiguous()):
            raise _lib.Y3Error("expected contiguous CUDA(HIP) tensors")
        if packed.dtype != torch.int32 or num_valid.dtype != torch.int32 or packed.dim() != 3 or packed.shape[2] != 7:
            raise _lib.Y3Error("packed must be int32 [b,M,7] and num_valid int32 [b]")
        b, m = packed.shape[0], packed.shape[1]
        if out is None:
            out = (torch.empty((self.world * b, m, 7), dtype=torch.int32, device=packed.device),
                   torch.empty((self.world * b,), dtype=torch.int32, device=packed.device))
        _lib.check(self.lib.y3_allgather_results(self._h, C.c_void_p(packed.data_ptr()), C.c_void_p(num_valid.data_ptr()),
                                                 b, m, C.c_void_p(out[0].data_ptr()), C.c_void_p(out[1].data_ptr()),
                                                 _lib.stream_ptr()), "y3_allgather_results")
        return out

    def close(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            self.lib.y3_comm_destroy(h)
            self._h = None

    __del__ = close


def allgather_ragged(packed: torch.Tensor, num_valid: torch.Tensor, n_images: int, group=None):
    """Uneven shards (n_images % world != 0): pad every rank's rows to the largest shard, gather, and drop
    the padding so that image order equals the un-sharded order."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return packed, num_valid
    world = dist.get_world_size(group)
    bmax = -(-n_images // world)
    pad = bmax - packed.shape[0]
    if pad:
        packed = torch.cat([packed, packed.new_zeros((pad,) + tuple(packed.shape[1:]))])
        num_valid = torch.cat([num_valid, num_valid.new_zeros((pad,))])
    g, nv = allgather_detections(packed, num_valid, group)
    keep = []
    for r in range(world):
        s, e = shard_range(n_images, r, world)
        keep.extend(range(r * bmax, r * bmax + (e - s)))
    idx = torch.tensor(keep, device=g.device, dtype=torch.long)
    return g.index_select(0, idx), nv.index_select(0, idx)
