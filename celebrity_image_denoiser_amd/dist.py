"""Multi-GPU use of the forward: one process per GPU, batch sharded, weights broadcast once.

Every image is independent in the forward (no batch-norm or cross-sample op; reference
backend/app.py:80-103), so the batch dimension is split contiguously over the ranks and the
forward needs no communication.  The only collective is one broadcast of the packed weights blob
(cid_packed_weights_bytes(), ~30 MB: every kernel layout plus a reference-layout copy) from the rank that loaded the checkpoint:
`torch.distributed.broadcast` on the "nccl" backend = RCCL over xGMI.  The reference has no
distributed code; this is the build's own data-parallel driver.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist

from . import _lib
from .generator import DenoiseGenerator


def shard_range(n_items: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous [begin, end) of rank `rank` when n_items are split over world_size ranks; the first
    n_items % world_size ranks take one extra item."""
    if world_size < 1 or not (0 <= rank < world_size) or n_items < 0:
        raise ValueError("bad shard arguments")
    q, r = divmod(n_items, world_size)
    begin = rank * q + min(rank, r)
    return begin, begin + q + (1 if rank < r else 0)


def broadcast_weights(model: DenoiseGenerator, src: int = 0, group: Optional[dist.ProcessGroup] = None) -> None:
    """Give every rank the weights of rank `src` with ONE broadcast of the packed blob.

    GPU ranks (backend nccl = RCCL): the device blob is broadcast and attached in place, and the
    nn.Parameters are refreshed from it.  CPU ranks (gloo, used by the CPU test-suite): the same
    bytes travel as a host tensor."""
    if not dist.is_initialized():
        raise RuntimeError("torch.distributed is not initialised")
    rank = dist.get_rank(group)
    on_gpu = next(model.parameters()).device.type == "cuda"
    nbytes = _lib.lib().cid_packed_weights_bytes()
    if rank == src:
        blob = model.pack_weights() if on_gpu else model.pack_weights_host()
    else:
        blob = torch.empty(nbytes, dtype=torch.uint8, device=next(model.parameters()).device)
    dist.broadcast(blob, src=src, group=group)
    if rank != src:
        model.adopt_packed_weights(blob, update_parameters=True)


def denoise_sharded(model: DenoiseGenerator, make_shard, n_items: int, group: Optional[dist.ProcessGroup] = None):
    """Run this rank's contiguous shard: `make_shard(begin, end)` returns the [end-begin,3,H,W] device
    tensor of those images; returns (begin, end, output).  No collective: outputs stay sharded."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    begin, end = shard_range(n_items, rank, world)
    if end == begin:
        return begin, end, None
    return begin, end, model(make_shard(begin, end))
