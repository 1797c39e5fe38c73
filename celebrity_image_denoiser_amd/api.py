"""model-load / denoise(image_batch) -> image_batch convenience layer.

Restates, for the build's own class, what the reference does around its module:
  * load_state_safely            reference backend/app.py:257-274 (+ the fallback at :327-336)
  * the value contract           reference backend/app.py:401-406 (input (u8/255-0.5)/0.5), :434-435
  * iterated denoising           reference backend/trainingcode/denoise_gan_code/denoise_eavl_iter.py:93-96
"""
from __future__ import annotations

import logging
from typing import Mapping, Optional, Sequence, Union

import torch

from .generator import DenoiseGenerator

logger = logging.getLogger("cid")


def extract_state_dict(ckpt, key_candidates: Sequence[str] = ("generator", "state_dict", "G")) -> Mapping:
    """Checkpoint object -> flat state_dict, like the reference loader: if the checkpoint is a dict
    holding a dict under one of `key_candidates` (the trainer saves {"generator": sd, ...},
    training.py:359-361) use that, else the object itself; then drop a leading "module." from
    every key (DataParallel checkpoints)."""
    state = ckpt
    if isinstance(ckpt, dict):
        for k in key_candidates:
            if k in ckpt and isinstance(ckpt[k], dict):
                state = ckpt[k]
                break
    out = {}
    for k, v in state.items():
        if isinstance(k, str) and k.startswith("module."):
            k = k.replace("module.", "")
        out[k] = v
    return out


def _numpy_scalar_globals():
    """What a pickled numpy scalar refers to (numpy 2 and numpy 1 spellings): the reconstruct function, `numpy.dtype` and the
    concrete dtype classes of the numeric kinds a metric can have.  Data constructors only — none of them runs file-supplied code."""
    import numpy as np

    try:
        from numpy._core.multiarray import scalar
    except ImportError:   # numpy < 2
        from numpy.core.multiarray import scalar
    kinds = ("float64", "float32", "float16", "int64", "int32", "int16", "int8", "uint64", "uint32", "uint16", "uint8", "bool")
    return [scalar, np.dtype] + sorted({type(np.dtype(k)) for k in kinds}, key=lambda t: t.__name__)


def _read_checkpoint_file(path: str) -> Mapping:
    """Checkpoint file -> flat state_dict of CPU tensors.

    The torch-free reader (ckpt.py: closed unpickling allow-list, nothing in the file can execute) handles the zip
    format every torch >= 1.6 writes.  Only when the file is NOT such an archive (the pre-1.6 legacy stream, for which
    the reader raises its explicit ValueError / zipfile.BadZipFile) does torch.load take over — with
    weights_only=True, so no pickle in the file gets to run code either way.  Every other failure (a checkpoint that
    references a disallowed global, a truncated storage, out-of-range strides) propagates: it is a bad file, not a
    reason to try a more permissive loader."""
    import zipfile

    from .ckpt import read_state_dict

    try:
        return {k: torch.from_numpy(v) for k, v in read_state_dict(path).items()}
    except (ValueError, zipfile.BadZipFile) as e:
        logger.info("%s is not a zip checkpoint (%s): reading the legacy format with torch.load(weights_only=True)", path, e)
        with torch.serialization.safe_globals(_numpy_scalar_globals()):
            return extract_state_dict(torch.load(path, map_location="cpu", weights_only=True))


def load_state_safely(model: torch.nn.Module, checkpoint_path: str,
                      key_candidates: Sequence[str] = ("generator", "state_dict", "G")) -> None:
    """torch.load -> extract_state_dict -> load_state_dict(strict=False) -> eval()  (app.py:257-274).
    weights_only=True with the numpy-scalar globals allow-listed: besides tensors and plain containers the trainer's
    checkpoint holds np.float64 values (`best_psnr`, `metric_history` = np.mean(...) results, training.py:368-369,449-465),
    which is why the reference itself reads it with weights_only=False (app.py:258); nothing else is let through."""
    with torch.serialization.safe_globals(_numpy_scalar_globals()):
        ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
    model.load_state_dict(extract_state_dict(ckpt, key_candidates), strict=False)
    model.eval()
    logger.info("Loaded PyTorch weights from %s", checkpoint_path)


def load(source: Union[str, Mapping, None] = None, device: Optional[Union[str, torch.device]] = None,
         strict: bool = False) -> DenoiseGenerator:
    """Build a DenoiseGenerator on `device` (default: current GPU) and load weights from a checkpoint
    path, a checkpoint dict or a state_dict.  `source=None` keeps the random initialisation — the
    state the reference server runs in when its checkpoint is missing (app.py:333-336)."""
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
    if device is None or torch.device(device).type != "cuda":
        raise RuntimeError("celebrity_image_denoiser_amd.load: an AMD GPU is required (no CPU fallback)")
    model = DenoiseGenerator()
    if isinstance(source, str):
        model.load_state_dict(_read_checkpoint_file(source), strict=strict)
    elif source is not None:
        sd = {k: (v if isinstance(v, torch.Tensor) else torch.as_tensor(v)) for k, v in extract_state_dict(source).items()}
        model.load_state_dict(sd, strict=strict)
    model.to(device).eval()
    model.pack_weights()
    return model


def denoise(model: DenoiseGenerator, image_batch: torch.Tensor, iterations: int = 1,
            max_batch: Optional[int] = None) -> torch.Tensor:
    """image_batch -> image_batch: fp32 [N,3,H,W] in [-1,1] -> fp32 [N,3,4*(H//4),4*(W//4)] in (-1,1).

    The batch may live on the host or the GPU; the result comes back where the input was.
    `iterations` feeds the output back in, on the device (denoise_eavl_iter.py:93-96);
    `max_batch` splits very large batches to bound the activation arena."""
    if iterations < 1:
        raise ValueError("iterations must be >= 1")
    dev = next(model.parameters()).device
    src_dev = image_batch.device
    outs = []
    step = max_batch or image_batch.shape[0]
    for i in range(0, image_batch.shape[0], step):
        x = image_batch[i:i + step].to(dev, torch.float32, non_blocking=True)
        for _ in range(iterations):
            x = model(x)
        outs.append(x)
    y = outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)
    return y.to(src_dev)


def denoise_u8(model: DenoiseGenerator, images_u8: torch.Tensor, max_batch: Optional[int] = None) -> torch.Tensor:
    """uint8 image batch [N,H,W,3] -> uint8 image batch [N,4*(H//4),4*(W//4),3]: the serving path of the
    reference (decode -> ToTensor -> Normalize -> net -> y*0.5+0.5 -> clamp -> ToPILImage, app.py:400-406,433-435,
    471-472) with everything between the two uint8 images on the GPU; 4x less PCIe traffic than fp32 tensors."""
    dev = next(model.parameters()).device
    src_dev = images_u8.device
    step = max_batch or images_u8.shape[0]
    outs = [model.forward_u8(images_u8[i:i + step].to(dev, non_blocking=True)) for i in range(0, images_u8.shape[0], step)]
    y = outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)
    return y.to(src_dev)


def get_padding(width: int, height: int, divisor: int = 4, scale: int = 1):
    """(left, top, right, bottom) that brings width/height up to a multiple of divisor*scale, split as evenly as
    integer halves allow with the extra pixel on the right/bottom — the reference's rule (app.py:276-281)."""
    d = divisor * scale
    pad_w = (d - width % d) % d
    pad_h = (d - height % d) % d
    return (pad_w // 2, pad_h // 2, pad_w - pad_w // 2, pad_h - pad_h // 2)


def serve_u8(model: DenoiseGenerator, images_u8: torch.Tensor, pad_divisor: int = 4) -> torch.Tensor:
    """The reference server's denoise path for images of ANY size (app.py:378-385, 400-406, 433-435, 471-480):
    pad the uint8 image with black to a multiple of `pad_divisor` (transforms.Pad(fill=0), i.e. -1 after
    normalisation), normalise, run the network, map back to uint8, crop the padding off again.
    images_u8: uint8 [N,H,W,3] (host or GPU) -> uint8 [N,H,W,3], same place.  The pad and the crop are index arithmetic inside
    the first and the last kernel (cid_forward_padded): no padded copy of the image and no uncropped result exist.  Only an
    image beyond one call's size limit (4,194,302 padded pixels) is padded in memory, because it is then cut into stripes."""
    if images_u8.dtype != torch.uint8 or images_u8.dim() != 4 or images_u8.shape[3] != 3:
        raise RuntimeError("serve_u8 expects a uint8 tensor of shape [N,H,W,3]")
    dev = next(model.parameters()).device
    src_dev = images_u8.device
    n, h, w, _ = images_u8.shape
    left, top, right, bottom = get_padding(w, h, pad_divisor)
    x = images_u8.to(dev, non_blocking=True)
    if model._needs_stripes(h + top + bottom, w + left + right):
        x = torch.nn.functional.pad(x, (0, 0, left, right, top, bottom), mode="constant", value=0)
        return model.forward_u8(x)[:, top:top + h, left:left + w, :].contiguous().to(src_dev)
    return model.forward_padded(x, (left, top, right, bottom), out_u8=True).to(src_dev)


def to_unit_range(y: torch.Tensor) -> torch.Tensor:
    """The reference's view transform for tanh-range outputs: y*0.5+0.5 clamped to [0,1] (app.py:435)."""
    return (y * 0.5 + 0.5).clamp(0, 1)
