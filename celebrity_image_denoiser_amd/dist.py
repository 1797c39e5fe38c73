"""Multi-GPU use of the forward: one process per GPU, batch sharded, weights broadcast once.

Every image is independent in the forward (no batch-norm or cross-sample op; reference
backend/app.py:80-103), so the batch dimension is split contiguously over the ranks and the
forward needs no communication.  The only collective is one broadcast of the packed weights blob
(cid_packed_weights_bytes(), ~26 MB: every kernel layout plus a reference-layout copy) from the rank that loaded the checkpoint:
`cid_broadcast_weights` = one `ncclBroadcast` (RCCL over xGMI) issued by libcid.so; torch.distributed carries only the 128-byte communicator id.  The reference has no
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


class WeightsComm:
    """An RCCL communicator over the ranks of a torch.distributed group, created through the C ABI
    (cid_comm_unique_id / cid_comm_init_rank, include/cid.h) so that the job's one collective — the broadcast of the
    packed weights — is an `ncclBroadcast` issued by libcid.so itself.  torch.distributed is only the control channel
    that carries the 128-byte unique id from rank `src` to the others."""

    def __init__(self, device: torch.device, group: Optional[dist.ProcessGroup] = None):
        import ctypes

        self._L = _lib.lib()
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        ident = torch.zeros(128, dtype=torch.uint8)
        if self.rank == 0:
            buf = (ctypes.c_char * 128)()
            _lib.check(None, self._L.cid_comm_unique_id(buf))
            ident = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
        box = [ident.tolist()]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        raw = bytes(box[0])
        self._comm = ctypes.c_void_p()
        with torch.cuda.device(device):
            _lib.check(None, self._L.cid_comm_init_rank(ctypes.byref(self._comm), self.world, raw, self.rank))

    def close(self) -> None:
        if self._comm:
            self._L.cid_comm_destroy(self._comm)
            self._comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def broadcast_weights(model: DenoiseGenerator, src: int = 0, group: Optional[dist.ProcessGroup] = None) -> None:
    """Give every rank the weights of rank `src` with ONE broadcast of the packed blob.

    GPU ranks: `cid_broadcast_weights` — one in-place `ncclBroadcast` (RCCL over xGMI) of the device blob, issued from
    the C ABI on the current stream; receivers attach it and refresh their nn.Parameters from it.  CPU ranks (gloo,
    the CPU test-suite): the same bytes travel as a host tensor through torch.distributed."""
    if not dist.is_initialized():
        raise RuntimeError("torch.distributed is not initialised")
    rank = dist.get_rank(group)
    dev = next(model.parameters()).device
    on_gpu = dev.type == "cuda"
    L = _lib.lib()
    nbytes = L.cid_packed_weights_bytes()
    if not on_gpu:
        blob = model.pack_weights_host() if rank == src else torch.empty(nbytes, dtype=torch.uint8)
        dist.broadcast(blob, src=src, group=group)
        if rank != src:
            model.adopt_packed_weights(blob, update_parameters=True)
        return
    # Transport 1: ncclBroadcast issued by libcid.so on its own communicator (cid_broadcast_weights).  Transport 2, only if
    # the first cannot be set up on this host (no RCCL found at run time, communicator creation refused): the same bytes as
    # ONE torch.distributed.broadcast on the process group's backend (nccl = the same RCCL over xGMI).  Either way it is one
    # collective of the packed blob; every rank takes the same branch (the choice is agreed on with an all-reduce).
    import logging

    comm, ok = None, 1
    try:
        comm = WeightsComm(dev, group)
    except Exception as e:   # noqa: BLE001 - any set-up failure selects transport 2 on ALL ranks
        logging.getLogger("cid").warning("cid_comm_* unavailable (%s): broadcasting the blob with torch.distributed", e)
        ok = 0
    flag = torch.tensor([ok], dtype=torch.int32, device=dev if dist.get_backend(group) == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    if int(flag.item()) == 0:
        if comm is not None:
            comm.close()
        blob = model.pack_weights() if rank == src else torch.empty(nbytes, dtype=torch.uint8, device=dev)
        dist.broadcast(blob, src=src, group=group)
        if rank != src:
            model.adopt_packed_weights(blob, update_parameters=True)
        return
    try:
        if rank == src:
            blob = model.pack_weights()
        else:
            blob = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            _lib.check(model._cid, L.cid_attach_weights(model._cid, blob.data_ptr()))
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            _lib.check(model._cid, L.cid_broadcast_weights(model._cid, comm._comm, src, rank, stream))
        if rank != src:
            model.adopt_packed_weights(blob, update_parameters=True, host_is_current=True)
        torch.cuda.current_stream(dev).synchronize()
    finally:
        comm.close()


def denoise_sharded(model: DenoiseGenerator, make_shard, n_items: int, group: Optional[dist.ProcessGroup] = None):
    """Run this rank's contiguous shard: `make_shard(begin, end)` returns the [end-begin,3,H,W] device
    tensor of those images; returns (begin, end, output).  No collective: outputs stay sharded."""
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    begin, end = shard_range(n_items, rank, world)
    if end == begin:
        return begin, end, None
    return begin, end, model(make_shard(begin, end))
