"""Host batches in, host batches out, with the PCIe copies hidden under the forward.

The reference moves one image per request to the device, runs the net and moves the result back, all on one
thread and one stream (backend/app.py:406,433-435; denoisegan_eval.py:91-99).  For a stream of host batches that
serialises H2D -> forward -> D2H; here the three run on three HIP streams over `depth` buffer slots, so that while
batch k computes, batch k+1 is uploading and batch k-1 is downloading.  The forward stays MFMA-bound (12 ms per
256 images), the copies are 12.6 MB (uint8) or 50 MB (fp32) each way: fully hidden.

    pipe = HostPipeline(model)                       # model: DenoiseGenerator on a GPU
    for out in pipe.run(batches):                    # uint8 [N,H,W,3] or float32 [N,3,H,W], numpy or CPU tensors
        ...                                          # out: CPU tensor, same format; results come in input order

PyTorch supplies streams, events and pinned memory (plumbing); the arithmetic is cid_forward / cid_forward_ex.
"""
from __future__ import annotations

from typing import Iterable, Iterator, List, Optional

import numpy as np
import torch

from .generator import DenoiseGenerator


class _Slot:
    __slots__ = ("key", "pin_in", "dev_in", "dev_out", "pin_out", "in_done", "fw_done", "out_done", "n", "busy", "src")

    def __init__(self):
        self.key = None
        self.busy = False
        self.n = 0
        self.src = None
        self.in_done = torch.cuda.Event()
        self.fw_done = torch.cuda.Event()
        self.out_done = torch.cuda.Event()


class HostPipeline:
    """Three-stream (upload / forward / download) pipeline over `depth` buffer slots.

    One forward is in flight per model at any time (the C handle owns one activation arena), which is what the
    single compute stream guarantees; `depth` only bounds how far uploads may run ahead (2 is enough to hide both
    copies, more only costs pinned memory)."""

    def __init__(self, model: DenoiseGenerator, depth: int = 2):
        if depth < 2:
            raise ValueError("depth must be >= 2 (one slot computing, one slot copying)")
        self.model = model
        self.dev = next(model.parameters()).device
        if self.dev.type != "cuda":
            raise RuntimeError("HostPipeline needs the model on an AMD GPU (no CPU fallback)")
        self.depth = depth
        with torch.cuda.device(self.dev):
            self.s_in, self.s_fw, self.s_out = (torch.cuda.Stream(self.dev) for _ in range(3))
            self.slots: List[_Slot] = [_Slot() for _ in range(depth)]

    # ------------------------------------------------------------------ buffers
    def _fit(self, slot: _Slot, shape, dtype) -> None:
        """(Re)allocate the slot for batches up to `shape` of `dtype`; smaller batches of the same image size reuse it."""
        key = (tuple(shape[1:]), dtype)
        if slot.key == key and slot.pin_in.shape[0] >= shape[0]:
            return
        n = shape[0]
        if dtype == torch.uint8:
            _, h, w, _ = shape
            oshape = (n, 4 * (h // 4), 4 * (w // 4), 3)
        else:
            _, _, h, w = shape
            oshape = (n, 3, 4 * (h // 4), 4 * (w // 4))
        if h < 4 or w < 4:
            raise RuntimeError(f"Given input size: ({h}x{w}). Calculated output size is too small (H and W must be >= 4)")
        slot.pin_in = torch.empty(tuple(shape), dtype=dtype).pin_memory()
        slot.pin_out = torch.empty(oshape, dtype=dtype).pin_memory()
        slot.dev_in = torch.empty(tuple(shape), dtype=dtype, device=self.dev)
        slot.dev_out = torch.empty(oshape, dtype=dtype, device=self.dev)
        slot.key = key

    @staticmethod
    def _as_tensor(batch) -> torch.Tensor:
        t = torch.from_numpy(np.ascontiguousarray(batch)) if isinstance(batch, np.ndarray) else batch
        if not isinstance(t, torch.Tensor) or t.device.type != "cpu":
            raise RuntimeError("HostPipeline takes host batches (numpy arrays or CPU tensors)")
        if t.dtype == torch.uint8 and t.dim() == 4 and t.shape[3] == 3:
            return t.contiguous()
        if t.dtype == torch.float32 and t.dim() == 4 and t.shape[1] == 3:
            return t.contiguous()
        raise RuntimeError(f"expected uint8 [N,H,W,3] or float32 [N,3,H,W], got {t.dtype} {list(t.shape)}")

    # ------------------------------------------------------------------ stages
    def _submit(self, slot: _Slot, t: torch.Tensor, iterations: int) -> None:
        n = t.shape[0]
        if n < 1:
            raise RuntimeError("empty batch")
        if slot.key != (tuple(t.shape[1:]), t.dtype) or slot.pin_in.shape[0] < n:
            torch.cuda.synchronize(self.dev)      # a shape change drains the pipeline before buffers are replaced
            self._fit(slot, t.shape, t.dtype)
        slot.n = n
        if t.is_pinned():
            slot.src = t                                          # already page-locked: upload straight from it (kept alive here)
        else:
            slot.pin_in[:n].copy_(t)                              # host memcpy into pinned memory
            slot.src = slot.pin_in[:n]
        u8 = t.dtype == torch.uint8
        with torch.cuda.device(self.dev):
            self.s_in.wait_event(slot.fw_done)                    # the previous forward on this slot has read dev_in
            with torch.cuda.stream(self.s_in):
                slot.dev_in[:n].copy_(slot.src, non_blocking=True)
                slot.in_done.record(self.s_in)
            self.s_fw.wait_event(slot.in_done)
            self.s_fw.wait_event(slot.out_done)                   # the previous download of dev_out has finished
            with torch.cuda.stream(self.s_fw):
                x, y = slot.dev_in[:n], slot.dev_out[:n]
                # iterations > 1 feed the fp32 output back in (denoise_eavl_iter.py:93-96); uint8 only at the two ends
                z = x
                for _ in range(iterations - 1):
                    z = self.model.forward_fmt(z, out_u8=False)
                self.model.forward_fmt(z, out_u8=u8, out=y)
                slot.fw_done.record(self.s_fw)
            self.s_out.wait_event(slot.fw_done)
            with torch.cuda.stream(self.s_out):
                slot.pin_out[:n].copy_(slot.dev_out[:n], non_blocking=True)
                slot.out_done.record(self.s_out)
        slot.busy = True

    def _collect(self, slot: _Slot, copy: bool) -> torch.Tensor:
        slot.out_done.synchronize()
        slot.busy = False
        out = slot.pin_out[:slot.n]
        return out.clone() if copy else out

    # ------------------------------------------------------------------ API
    def run(self, batches: Iterable, iterations: int = 1, copy: bool = True) -> Iterator[torch.Tensor]:
        """Yield the denoised batch for every host batch, in order.  With copy=False the yielded tensor is a view
        of a pinned slot buffer, valid only until the generator is advanced again (its slot is then refilled)."""
        if iterations < 1:
            raise ValueError("iterations must be >= 1")
        pending: List[_Slot] = []
        k = 0
        # the model's arena is shared with eager calls on the caller's stream: order this run's forwards behind them
        self.s_fw.wait_stream(torch.cuda.current_stream(self.dev))
        try:
            for batch in batches:
                slot = self.slots[k % self.depth]
                if slot.busy:                       # oldest result still sits in this slot: hand it out first
                    assert pending and pending[0] is slot
                    yield self._collect(pending.pop(0), copy)
                self._submit(slot, self._as_tensor(batch), iterations)
                pending.append(slot)
                k += 1
            while pending:
                yield self._collect(pending.pop(0), copy)
        finally:
            torch.cuda.synchronize(self.dev)
            for s in self.slots:
                s.busy = False

    def __call__(self, batches: Iterable, iterations: int = 1) -> List[torch.Tensor]:
        return list(self.run(batches, iterations=iterations, copy=True))


def denoise_host_batches(model: DenoiseGenerator, batches: Iterable, iterations: int = 1,
                         depth: int = 2) -> Optional[List[torch.Tensor]]:
    """Convenience: run a list/iterator of host batches through a HostPipeline and return the list of results."""
    return HostPipeline(model, depth=depth)(batches, iterations=iterations)


class GraphedForward:
    """One forward at a fixed shape, captured into a HIP graph: a replay is ONE launch instead of twelve.

    The reference serves one image per request (backend/app.py:406,433): at N=1 the twelve kernel launches of a forward
    are short, and the host-side launch path is a visible part of the latency.  `cid_forward` only enqueues kernels (no
    allocation, no synchronisation), so it can be stream-captured once and replayed.

        fast = GraphedForward(model, example)        # example: fp32 [N,3,H,W] or uint8 [N,H,W,3] on the GPU
        y = fast(x)                                  # same shape/dtype as example; result as `model(x)` / `forward_u8(x)`

    The result tensor is owned by the graph and overwritten by the next call (clone it to keep it)."""

    def __init__(self, model: DenoiseGenerator, example: torch.Tensor):
        if example.device.type != "cuda":
            raise RuntimeError("GraphedForward needs a GPU tensor as the example (no CPU fallback)")
        self.model = model
        self.u8 = example.dtype == torch.uint8
        self.static_in = example.clone().contiguous()
        self._capture()

    def _capture(self) -> None:
        """Capture one forward.  The graph bakes in raw device addresses, so everything it touches is owned here: the
        input/output tensors, a PRIVATE activation arena (the model's own may be replaced when a later eager call needs a
        bigger one) and a reference to the packed weights blob it was captured with (a parameter update makes the model
        pack a new blob; `__call__` notices and re-captures)."""
        model, dev = self.model, self.static_in.device
        self._blob = model.pack_weights()
        self._sig = model._packed_sig
        shape = self.static_in.shape
        n, h, w = (shape[0], shape[1], shape[2]) if self.u8 else (shape[0], shape[2], shape[3])
        if model._needs_stripes(h, w):
            raise RuntimeError("GraphedForward: images beyond the single-call size limit run as several striped forwards "
                               "with intermediate allocations; call the model directly")
        shared, model._ws = model._ws, None
        try:
            model._ensure_arena(n, h, w, dev)
            self._ws = model._ws                                     # the graph's own arena
            model.forward_fmt(self.static_in, out_u8=self.u8)        # warm: nothing allocates inside the capture
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                with torch.cuda.graph(self.graph, stream=side):
                    self.static_out = model.forward_fmt(self.static_in, out_u8=self.u8)
            torch.cuda.current_stream(dev).wait_stream(side)
            assert model._ws is self._ws and model._blob is self._blob
        finally:
            model._ws = shared

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        if x.shape != self.static_in.shape or x.dtype != self.static_in.dtype or x.device != self.static_in.device:
            raise RuntimeError(f"GraphedForward was captured for {self.static_in.dtype} {list(self.static_in.shape)} on "
                               f"{self.static_in.device}, got {x.dtype} {list(x.shape)} on {x.device}")
        if self.model.pack_weights() is not self._blob or self.model._packed_sig != self._sig:
            self._capture()                                          # the weights changed since the capture
        self.static_in.copy_(x, non_blocking=True)
        self.graph.replay()
        return self.static_out
