"""`DenoiseGenerator`: the reference's nn.Module surface over the HIP forward.

The reference's boundary for this path is the torch.nn.Module protocol on the object stored in
`PT_MODELS["denoise"]` (reference backend/app.py:319-320); its callers use exactly
    DenoiseGenerator()                          app.py:320, denoisegan_eval.py:67
    .to(device)                                 app.py:320
    .load_state_dict(sd, strict=False|True)     app.py:272, denoisegan_eval.py:69
    .eval() -> self                             app.py:273,423
    __call__(x)                                 app.py:433, denoise_eavl_iter.py:96
    .state_dict()                               training.py:361
This class offers the same calls with the same parameter names, shapes and default
initialisation (the submodules below are parameter containers built from stock nn layers, so
key names, `.to()`, `state_dict()` and `load_state_dict()` behave as in the reference), but
`forward` never touches ATen compute: it hands raw device pointers to `cid_forward`
(include/cid.h), which runs the hand-written gfx950 kernels.  There is no CPU path — a CPU
tensor or a missing libcid.so raises.
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch
import torch.nn as nn

from . import _lib


def _block(cin, cmid, cout, last_relu=True):
    layers = [nn.Conv2d(cin, cmid, kernel_size=3, padding=1), nn.ReLU(), nn.Conv2d(cmid, cout, kernel_size=3, padding=1)]
    if last_relu:
        layers.append(nn.ReLU())
    return nn.Sequential(*layers)


class DenoiseGenerator(nn.Module):
    """Two-level U-Net denoiser; parameters as in reference backend/app.py:39-78."""

    def __init__(self):
        super().__init__()
        # parameter containers only — indices 0 and 2 of each Sequential hold the convs, as in the reference
        self.down1 = _block(3, 64, 64)
        self.pool1 = nn.MaxPool2d(2, 2)
        self.down2 = _block(64, 128, 128)
        self.pool2 = nn.MaxPool2d(2, 2)
        self.bottleneck = _block(128, 256, 256)
        self.up2 = nn.ConvTranspose2d(256, 128, kernel_size=2, stride=2)
        self.upconv2 = _block(256, 128, 128)
        self.up1 = nn.ConvTranspose2d(128, 64, kernel_size=2, stride=2)
        self.upconv1 = _block(128, 64, 3, last_relu=False)
        self._cid = ctypes.c_void_p()
        _lib.check(None, _lib.lib().cid_create(ctypes.byref(self._cid)))
        self._blob = None          # packed weights on the device (torch uint8 tensor, owns the memory)
        self._packed_sig = None    # signature of the parameters the blob was packed from
        self._ws = None            # activation arena (torch uint8 tensor, grow-only)

    def __del__(self):
        try:
            if getattr(self, "_cid", None):
                _lib.lib().cid_destroy(self._cid)
                self._cid = None
        except Exception:
            pass

    # ------------------------------------------------------------------ weights
    def _signature(self):
        # p._version counts in-place updates made through the tensor API; writes through `p.data` or raw pointers do not
        # bump it — after such a write call pack_weights(force=True).
        return tuple((k, p.data_ptr(), p._version, str(p.device)) for k, p in self.named_parameters())

    def _device(self) -> torch.device:
        return next(self.parameters()).device

    def _stage_parameters(self) -> None:
        L = _lib.lib()
        for key, p in self.named_parameters():
            a = np.ascontiguousarray(p.detach().to("cpu", torch.float32).numpy())
            shape = (ctypes.c_int64 * a.ndim)(*a.shape)
            _lib.check(self._cid, L.cid_set_weight(self._cid, key.encode(), a.ctypes.data, shape, a.ndim))

    def pack_weights_host(self) -> torch.Tensor:
        """The packed weights blob as a host uint8 tensor (what `pack_weights` uploads): lets a
        transport that is not RCCL (gloo in the CPU tests) move the same bytes."""
        L = _lib.lib()
        self._stage_parameters()
        out = np.empty(L.cid_packed_weights_bytes(), dtype=np.uint8)
        _lib.check(self._cid, L.cid_export_packed(self._cid, out.ctypes.data, out.nbytes))
        return torch.from_numpy(out)

    def pack_weights(self, force: bool = False) -> torch.Tensor:
        """Repack the 24 parameter tensors into the kernels' layout on the module's GPU (if they
        changed since the last call) and return the packed device blob."""
        sig = self._signature()
        if not force and self._blob is not None and sig == self._packed_sig:
            return self._blob
        dev = self._device()
        if dev.type != "cuda":
            raise RuntimeError(
                "DenoiseGenerator runs only on an AMD GPU (HIP kernels behind libcid.so); move it with "
                ".to('cuda') first. There is no CPU fallback."
            )
        L = _lib.lib()
        self._stage_parameters()
        blob = torch.empty(L.cid_packed_weights_bytes(), dtype=torch.uint8, device=dev)
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            _lib.check(self._cid, L.cid_upload_weights(self._cid, blob.data_ptr(), stream))
        self._blob, self._packed_sig = blob, sig
        return blob

    def adopt_packed_weights(self, blob: torch.Tensor, update_parameters: bool = True, host_is_current: bool = False) -> None:
        """Use a packed device blob produced elsewhere (another rank's `pack_weights()`, received
        by RCCL broadcast).  With update_parameters the nn.Parameters are refreshed from it so
        `state_dict()` agrees with what the kernels compute.  host_is_current: the handle's host copy already
        holds these bytes (cid_broadcast_weights refreshes it on receivers) — skip the device-to-host copy."""
        L = _lib.lib()
        if blob.dtype != torch.uint8 or blob.numel() != L.cid_packed_weights_bytes() or not blob.is_contiguous():
            raise ValueError("adopt_packed_weights: expected a contiguous uint8 tensor of cid_packed_weights_bytes()")
        if blob.device.type == "cuda":
            if self._device() != blob.device:
                self.to(blob.device)
            _lib.check(self._cid, L.cid_attach_weights(self._cid, blob.data_ptr()))
            self._blob = blob
        else:
            update_parameters = True   # a host blob can only refresh the parameters; packing happens on .to('cuda')
        if update_parameters:
            if not host_is_current:
                host = blob.cpu().numpy()
                _lib.check(self._cid, L.cid_import_packed(self._cid, host.ctypes.data, host.nbytes))
            with torch.no_grad():
                for key, p in self.named_parameters():
                    a = np.empty(tuple(p.shape), dtype=np.float32)
                    _lib.check(self._cid, L.cid_get_weight(self._cid, key.encode(), a.ctypes.data, a.size))
                    p.copy_(torch.from_numpy(a))
        self._packed_sig = self._signature() if blob.device.type == "cuda" else None

    # ------------------------------------------------------------------ forward
    def _prepare(self, x: torch.Tensor, out: torch.Tensor = None):
        if not isinstance(x, torch.Tensor):
            raise TypeError("DenoiseGenerator expects a torch.Tensor [N,3,H,W]")
        if x.dim() != 4 or x.shape[1] != 3:
            raise RuntimeError(f"expected input of shape [N,3,H,W], got {list(x.shape)}")
        if x.device.type != "cuda":
            raise RuntimeError(
                "DenoiseGenerator.forward got a CPU tensor: this implementation is GPU-only (hand-written HIP "
                "kernels); there is no CPU fallback. Move the input with .to('cuda')."
            )
        if x.dtype != torch.float32:
            raise RuntimeError(f"expected float32 input (the reference computes in fp32), got {x.dtype}")
        if x.device != self._device():
            raise RuntimeError(f"input on {x.device} but module parameters on {self._device()}")
        n, _, h, w = x.shape
        if n < 1:
            raise RuntimeError("empty batch")
        L = _lib.lib()
        ho, wo = ctypes.c_int(), ctypes.c_int()
        if L.cid_out_shape(h, w, ctypes.byref(ho), ctypes.byref(wo)) != _lib.CID_OK:
            # the reference fails here too (ATen: "Output size is too small")
            raise RuntimeError(f"Given input size: ({h}x{w}). Calculated output size is too small (H and W must be >= 4)")
        self.pack_weights()
        self._ensure_arena(n, h, w, x.device)
        x = x.contiguous()
        y = self._output(out, (n, 3, ho.value, wo.value), torch.float32, x.device)
        return x, y, n, h, w

    def _ensure_arena(self, n: int, h: int, w: int, device: torch.device) -> None:
        """Grow-only activation arena.  Replacing it waits for the device first: kernels of an earlier forward (possibly on
        another stream, e.g. HostPipeline's compute stream) may still be using the old one, and the caching allocator
        would hand its memory to the next allocation of the stream that owns it."""
        need = ctypes.c_size_t()
        _lib.check(self._cid, _lib.lib().cid_workspace_bytes(n, h, w, ctypes.byref(need)))
        if self._ws is None or self._ws.numel() < need.value or self._ws.device != device:
            if self._ws is not None:
                torch.cuda.synchronize(self._ws.device)
            self._ws = None
            self._ws = torch.empty(need.value, dtype=torch.uint8, device=device)

    @staticmethod
    def _output(out, shape, dtype, device) -> torch.Tensor:
        """A fresh output tensor, or the caller's `out` after checking it is exactly what the kernels will write."""
        if out is None:
            return torch.empty(shape, dtype=dtype, device=device)
        if tuple(out.shape) != tuple(shape) or out.dtype != dtype or out.device != device or not out.is_contiguous():
            raise RuntimeError(f"out= must be a contiguous {dtype} tensor of shape {list(shape)} on {device}, "
                               f"got {out.dtype} {list(out.shape)} on {out.device}")
        return out

    def forward(self, x: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
        """[N,3,H,W] fp32 in [-1,1] on the GPU -> [N,3,4*(H//4),4*(W//4)] fp32 in (-1,1).
        Same contract as the reference forward (app.py:80-103); asynchronous on the current stream.
        `out=` (not in the reference) writes into a caller-owned tensor instead of allocating."""
        if isinstance(x, torch.Tensor) and x.dim() == 4 and self._needs_stripes(x.shape[2], x.shape[3]):
            return self._forward_striped(x, out_u8=False, out=out)
        x, y, n, h, w = self._prepare(x, out)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        with torch.cuda.device(x.device):
            _lib.check(self._cid, _lib.lib().cid_forward(self._cid, x.data_ptr(), y.data_ptr(), n, h, w,
                                                         self._ws.data_ptr(), self._ws.numel(), stream))
        return y

    # ------------------------------------------------------------------ images beyond one call's size limit
    # The kernels address one image's activations with 32-bit byte offsets: cid_forward refuses H*W >= 4,194,303 pixels
    # (include/cid.h; 2048x2048 is the first square that does not fit).  The reference takes any size (app.py:80-103 is fully
    # convolutional), so larger images are cut into horizontal stripes here: output rows [a, b) (multiples of 8) are computed
    # from input rows [a - 32, b + 32).  A network output depends on input rows within +-20 (2 + 4 + 8 + 4 + 2 rows through
    # the three resolutions, plus pool alignment), stripe origins are multiples of 16 (pool windows, transposed-conv phases
    # and the 4x4 Winograd tiles keep their alignment down to the quarter-resolution layers: a pixel that changed its place
    # inside a Winograd tile would be summed in another order), and every pixel is computed by the same instructions, so
    # the assembled result equals the single-call result bit for bit (tests: forced small stripes on a mid-size image).
    STRIPE_HALO = 32
    MAX_PIXELS_PER_CALL = 0x7ffffff0 // 512 - 1

    def _needs_stripes(self, h: int, w: int) -> bool:
        return h * w > self.MAX_PIXELS_PER_CALL

    def _forward_striped(self, x: torch.Tensor, out_u8: bool, out: torch.Tensor = None, stripe_rows: int = None) -> torch.Tensor:
        in_u8 = x.dtype == torch.uint8
        n = x.shape[0]
        h, w = (x.shape[1], x.shape[2]) if in_u8 else (x.shape[2], x.shape[3])
        if h < 4 or w < 4:
            raise RuntimeError(f"Given input size: ({h}x{w}). Calculated output size is too small (H and W must be >= 4)")
        ho, wo = 4 * (h // 4), 4 * (w // 4)
        halo = self.STRIPE_HALO
        if stripe_rows is None:
            stripe_rows = (self.MAX_PIXELS_PER_CALL // w - 2 * halo) // 16 * 16
        if stripe_rows < 16 or stripe_rows % 16:
            raise RuntimeError(f"image rows of {w} pixels are too wide to cut into stripes of at least 16 rows")
        shape = (n, ho, wo, 3) if out_u8 else (n, 3, ho, wo)
        y = self._output(out, shape, torch.uint8 if out_u8 else torch.float32, x.device)
        for a in range(0, ho, stripe_rows):
            b = min(a + stripe_rows, ho)
            i0, i1 = max(0, a - halo), (h if b == ho else min(h, b + halo))
            xs = (x[:, i0:i1] if in_u8 else x[:, :, i0:i1]).contiguous()
            ys = self.forward_fmt(xs, out_u8=out_u8)
            if out_u8:
                y[:, a:b] = ys[:, a - i0:b - i0]
            else:
                y[:, :, a:b] = ys[:, :, a - i0:b - i0]
        return y

    def forward_u8(self, images: torch.Tensor, out_u8: bool = True, out: torch.Tensor = None) -> torch.Tensor:
        """uint8 images [N,H,W,3] (PIL/numpy layout) on the GPU -> denoised images, with the reference's
        pre/post-processing folded into the first/last kernel (cid_forward_ex):
        input  (u8/255 - 0.5)/0.5                       app.py:401-405 (ToTensor + Normalize(0.5, 0.5))
        output (uint8)(clamp(y*0.5+0.5, 0, 1) * 255)    app.py:435, 471-472 (ToPILImage truncates)
        Returns uint8 [N,4*(H//4),4*(W//4),3], or with out_u8=False the fp32 NCHW tensor `forward` returns."""
        if not isinstance(images, torch.Tensor) or images.dtype != torch.uint8 or images.dim() != 4 or images.shape[3] != 3:
            raise RuntimeError("forward_u8 expects a uint8 tensor of shape [N,H,W,3]")
        return self.forward_fmt(images, out_u8=out_u8, out=out)

    def forward_fmt(self, x: torch.Tensor, out_u8: bool, out: torch.Tensor = None) -> torch.Tensor:
        """The forward with either caller-side format on either side (cid_forward_ex): `x` is uint8 [N,H,W,3] or
        fp32 [N,3,H,W] (told apart by dtype), the result uint8 [N,Ho,Wo,3] (out_u8) or fp32 [N,3,Ho,Wo]."""
        if not isinstance(x, torch.Tensor) or x.dim() != 4:
            raise RuntimeError("expected a 4-d tensor: uint8 [N,H,W,3] or float32 [N,3,H,W]")
        in_u8 = x.dtype == torch.uint8
        if in_u8 and x.shape[3] != 3:
            raise RuntimeError(f"expected uint8 images of shape [N,H,W,3], got {list(x.shape)}")
        if not in_u8 and (x.dtype != torch.float32 or x.shape[1] != 3):
            raise RuntimeError(f"expected float32 input of shape [N,3,H,W], got {x.dtype} {list(x.shape)}")
        if x.device.type != "cuda":
            raise RuntimeError("got a CPU tensor: this implementation is GPU-only; there is no CPU fallback")
        if x.device != self._device():
            raise RuntimeError(f"input on {x.device} but module parameters on {self._device()}")
        n = x.shape[0]
        h, w = (x.shape[1], x.shape[2]) if in_u8 else (x.shape[2], x.shape[3])
        if n >= 1 and self._needs_stripes(h, w):
            return self._forward_striped(x, out_u8=out_u8, out=out)
        L = _lib.lib()
        ho, wo = ctypes.c_int(), ctypes.c_int()
        if n < 1 or L.cid_out_shape(h, w, ctypes.byref(ho), ctypes.byref(wo)) != _lib.CID_OK:
            raise RuntimeError(f"Given input size: ({h}x{w}). Calculated output size is too small (H and W must be >= 4)")
        self.pack_weights()
        self._ensure_arena(n, h, w, x.device)
        x = x.contiguous()
        if out_u8:
            y = self._output(out, (n, ho.value, wo.value, 3), torch.uint8, x.device)
        else:
            y = self._output(out, (n, 3, ho.value, wo.value), torch.float32, x.device)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        with torch.cuda.device(x.device):
            _lib.check(self._cid, L.cid_forward_ex(self._cid, x.data_ptr(), _lib.CID_FMT_U8_NHWC if in_u8 else _lib.CID_FMT_F32_NCHW,
                                                   y.data_ptr(), _lib.CID_FMT_U8_NHWC if out_u8 else _lib.CID_FMT_F32_NCHW,
                                                   n, h, w, self._ws.data_ptr(), self._ws.numel(), stream))
        return y

    def forward_padded(self, x: torch.Tensor, padding, out_u8: bool, out: torch.Tensor = None) -> torch.Tensor:
        """The reference server's pad -> network -> crop for one image size (app.py:276-281,384-385,474-480) in ONE call
        (cid_forward_padded): `x` is the caller's UNPADDED batch, uint8 [N,H,W,3] or fp32 [N,3,H,W]; `padding` =
        (left, top, right, bottom) as get_padding returns it; the band is black (uint8 0 = -1.0 normalised).  The result has
        the caller's H x W: the padding never exists in memory on either side — the first kernel synthesises it, the last one
        skips it."""
        if not isinstance(x, torch.Tensor) or x.dim() != 4:
            raise RuntimeError("expected a 4-d tensor: uint8 [N,H,W,3] or float32 [N,3,H,W]")
        in_u8 = x.dtype == torch.uint8
        if (in_u8 and x.shape[3] != 3) or (not in_u8 and (x.dtype != torch.float32 or x.shape[1] != 3)):
            raise RuntimeError(f"expected uint8 [N,H,W,3] or float32 [N,3,H,W], got {x.dtype} {list(x.shape)}")
        if x.device.type != "cuda":
            raise RuntimeError("got a CPU tensor: this implementation is GPU-only; there is no CPU fallback")
        if x.device != self._device():
            raise RuntimeError(f"input on {x.device} but module parameters on {self._device()}")
        left, top, right, bottom = (int(v) for v in padding)
        n = x.shape[0]
        h, w = (x.shape[1], x.shape[2]) if in_u8 else (x.shape[2], x.shape[3])
        hp, wp = h + top + bottom, w + left + right
        if n < 1:
            raise RuntimeError("empty batch")
        if self._needs_stripes(hp, wp):
            raise RuntimeError(f"padded image {hp}x{wp} is beyond one call's size limit: pad on the host and use forward_fmt (stripes)")
        L = _lib.lib()
        self.pack_weights()
        self._ensure_arena(n, hp, wp, x.device)
        x = x.contiguous()
        y = self._output(out, (n, h, w, 3) if out_u8 else (n, 3, h, w), torch.uint8 if out_u8 else torch.float32, x.device)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        with torch.cuda.device(x.device):
            _lib.check(self._cid, L.cid_forward_padded(self._cid, x.data_ptr(), _lib.CID_FMT_U8_NHWC if in_u8 else _lib.CID_FMT_F32_NCHW,
                                                       y.data_ptr(), _lib.CID_FMT_U8_NHWC if out_u8 else _lib.CID_FMT_F32_NCHW,
                                                       n, h, w, left, top, right, bottom, self._ws.data_ptr(), self._ws.numel(), stream))
        return y

    @staticmethod
    def view_u8(y: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
        """The reference's image view of a tanh-range tensor (app.py:435,471-472: y*0.5+0.5, clamp, mul(255).byte()) as one HIP
        kernel (cid_view_u8): fp32 [N,3,H,W] on the GPU -> uint8 [N,H,W,3].  What `forward_fmt(out_u8=True)` would have stored,
        for callers that also keep `y` itself (the iterated denoise feeds it back, denoise_eavl_iter.py:93-110)."""
        if not isinstance(y, torch.Tensor) or y.dim() != 4 or y.shape[1] != 3 or y.dtype != torch.float32:
            raise RuntimeError("view_u8 expects a float32 tensor of shape [N,3,H,W]")
        if y.device.type != "cuda":
            raise RuntimeError("got a CPU tensor: this implementation is GPU-only; there is no CPU fallback")
        n, _, h, w = y.shape
        if n < 1 or h < 1 or w < 1:
            raise RuntimeError("empty tensor")
        y = y.contiguous()
        img = DenoiseGenerator._output(out, (n, h, w, 3), torch.uint8, y.device)
        stream = torch.cuda.current_stream(y.device).cuda_stream
        with torch.cuda.device(y.device):
            _lib.check(None, _lib.lib().cid_view_u8(y.data_ptr(), img.data_ptr(), n, h, w, stream))
        return img

    def forward_timed(self, x: torch.Tensor):
        """forward + per-launch milliseconds from HIP events on the launch stream (measurement aid)."""
        x, y, n, h, w = self._prepare(x)
        ms = (ctypes.c_float * _lib.CID_NUM_LAUNCHES)()
        stream = torch.cuda.current_stream(x.device).cuda_stream
        with torch.cuda.device(x.device):
            _lib.check(self._cid, _lib.lib().cid_forward_timed(self._cid, x.data_ptr(), y.data_ptr(), n, h, w,
                                                               self._ws.data_ptr(), self._ws.numel(), stream, ms))
        return y, list(ms)


    def stage_output(self, stage: str, n: int, h: int, w: int) -> torch.Tensor:
        """Testing aid: the output of the reference module's submodule `stage` ("down1", "pool1", "down2", "pool2",
        "bottleneck", "up2", "upconv2", "up1" — what a forward hook on it would record, app.py:81-96) as left in the
        activation arena by the LAST forward of an [n,3,h,w] batch, returned as a fresh fp32 NCHW tensor.  Skip tensors
        ("down1", "down2") cover only the top-left region the concat keeps (cid_stage_view, include/cid.h)."""
        L = _lib.lib()
        off, c, hs, ws, ps, coff = ctypes.c_size_t(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        rc = L.cid_stage_view(stage.encode(), n, h, w, ctypes.byref(off), ctypes.byref(c), ctypes.byref(hs), ctypes.byref(ws),
                              ctypes.byref(ps), ctypes.byref(coff))
        if rc != _lib.CID_OK:
            raise KeyError(f"no stored stage {stage!r} for input [{n},3,{h},{w}] (cid_stage_view -> {rc})")
        if self._ws is None:
            raise RuntimeError("stage_output: no forward has run yet")
        half = self.compute_dtype == "f16"
        flat = self._ws[off.value:].view(torch.float16 if half else torch.float32)
        view = flat[: n * hs.value * ws.value * ps.value].view(n, hs.value, ws.value, ps.value)[..., coff.value:coff.value + c.value]
        return view.permute(0, 3, 1, 2).to(torch.float32).contiguous()

    @property
    def conv_algo(self) -> str:
        """Algorithm of the eight 3x3 GEMM layers: "winograd42" (default; Winograd F(4x2,3x3) on MFMA, 3 multiplies per
        output pixel and channel pair), "winograd64" (Winograd F(2x2,3x3), 4 multiplies), "direct" (9-tap implicit GEMM, 9) — all three on the
        exact-fp32 MFMA — or, opt-in, "split16": the direct form with every fp32 operand taken as two halfs (hi + lo) and every product as
        three fp16-MFMA products with fp32 accumulation (include/cid.h, CID_ALGO_SPLIT16: the least accurate of the four — 1.2x the direct fp32 kernel's error,
        2-3x ATen fp32's — inside the same 1e-5 contract, and not plain fp32 arithmetic)."""
        a = ctypes.c_int()
        _lib.check(self._cid, _lib.lib().cid_get_conv_algo(self._cid, ctypes.byref(a)))
        return {_lib.CID_ALGO_WINOGRAD64: "winograd64", _lib.CID_ALGO_WINOGRAD42: "winograd42", _lib.CID_ALGO_SPLIT16: "split16"}.get(a.value, "direct")

    @conv_algo.setter
    def conv_algo(self, name: str) -> None:
        algo = {"direct": _lib.CID_ALGO_DIRECT, "winograd64": _lib.CID_ALGO_WINOGRAD64, "winograd42": _lib.CID_ALGO_WINOGRAD42, "split16": _lib.CID_ALGO_SPLIT16}.get(name)
        if algo is None:
            raise ValueError("conv_algo must be 'direct', 'winograd64', 'winograd42' or 'split16'")
        _lib.check(self._cid, _lib.lib().cid_set_conv_algo(self._cid, algo))

    @property
    def tail_algo(self) -> str:
        """"fused" (default: the last layer's channel contraction runs inside upconv1[0]'s kernel — Winograd or split16; with conv_algo
        "direct" or compute_dtype "f16" it behaves as "bands"), "bands" (separate row-band kernel, images up to 128 pixels
        wide) or "tiles" (round 1's tiled kernel).  Same function; all go through the parity tests."""
        a = ctypes.c_int()
        _lib.check(self._cid, _lib.lib().cid_get_tail_algo(self._cid, ctypes.byref(a)))
        return {_lib.CID_TAIL_TILES: "tiles", _lib.CID_TAIL_BANDS: "bands"}.get(a.value, "fused")

    @tail_algo.setter
    def tail_algo(self, name: str) -> None:
        algo = {"fused": _lib.CID_TAIL_FUSED, "bands": _lib.CID_TAIL_BANDS, "tiles": _lib.CID_TAIL_TILES}.get(name)
        if algo is None:
            raise ValueError("tail_algo must be 'fused', 'bands' or 'tiles'")
        _lib.check(self._cid, _lib.lib().cid_set_tail_algo(self._cid, algo))

    @property
    def compute_dtype(self) -> str:
        """"f32" (default: the reference's arithmetic) or "f16" (half storage between the first and last kernel,
        fp16 MFMA with fp32 accumulators; BASELINE configs[4]).  Inputs/outputs keep their formats."""
        d = ctypes.c_int()
        _lib.check(self._cid, _lib.lib().cid_get_compute_dtype(self._cid, ctypes.byref(d)))
        return "f16" if d.value == _lib.CID_DTYPE_F16 else "f32"

    @compute_dtype.setter
    def compute_dtype(self, name: str) -> None:
        d = {"f32": _lib.CID_DTYPE_F32, "f16": _lib.CID_DTYPE_F16}.get(name)
        if d is None:
            raise ValueError("compute_dtype must be 'f32' or 'f16'")
        _lib.check(self._cid, _lib.lib().cid_set_compute_dtype(self._cid, d))

    def timing_begin(self, max_forwards: int) -> None:
        """Arm per-launch HIP-event timing for the next `max_forwards` forwards (no per-forward sync)."""
        _lib.check(self._cid, _lib.lib().cid_timing_begin(self._cid, int(max_forwards)))

    def timing_end(self):
        """-> (per-launch milliseconds summed over the recorded forwards, number of forwards)."""
        ms = (ctypes.c_float * _lib.CID_NUM_LAUNCHES)()
        n = ctypes.c_int()
        dev = self._device()
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            _lib.check(self._cid, _lib.lib().cid_timing_end(self._cid, stream, ms, ctypes.byref(n)))
        return list(ms), n.value


def launch_table(n: int, h: int, w: int, model: "DenoiseGenerator" = None):
    """[(layer name, kernel symbol, algorithmic flops, algorithmic bytes, executed flops)] of one forward, per LAUNCH under
    `model`'s configuration (direct kernels, unfused, if no model is given).  Algorithmic = the direct-convolution count of
    the reference layer(s) the launch computes; executed = what its MFMAs issue: the Winograd kernels run 24/72 (F(4x2,3x3)) or 16/36 (F(2x2,3x3)) of
    their 3x3 layer's multiplies (a fused-in contraction of the next layer is executed as it stands); the split-operand kernels (conv_algo "split16")
    issue THREE fp16-MFMA products per multiply."""
    L = _lib.lib()
    rows = []
    handle = model._cid if model is not None else None
    for i in range(_lib.CID_NUM_LAUNCHES):
        f, b, f0, b0 = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        _lib.check(None, L.cid_launch_work_ex(handle, i, n, h, w, ctypes.byref(f), ctypes.byref(b)))
        _lib.check(None, L.cid_launch_work(i, n, h, w, ctypes.byref(f0), ctypes.byref(b0)))
        kern = L.cid_launch_kernel(handle, i).decode()
        own = min(f.value, f0.value)                       # the launch's own layer (0 for a launch that only sums)
        split = kern.startswith("k_conv3x3_h16<") and len(kern.split(",")) > 5 and kern.split(",")[5].strip().startswith("true")      # conv_algo "split16" (template flag F32IO): three fp16-MFMA products per multiply
        executed = own * (3.0 if split else 24.0 / 72.0 if "wino42" in kern else 16.0 / 36.0 if "wino" in kern else 1.0) + (f.value - own)
        rows.append((L.cid_launch_name(i).decode(), kern, f.value, b.value, executed))
    return rows
