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
