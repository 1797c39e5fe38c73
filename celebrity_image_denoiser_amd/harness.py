"""Directory-to-directory batch denoising: the reference's offline eval harnesses on the GPU pipeline.

Mirrors `enhance_images` of reference backend/trainingcode/denoise_gan_code/denoisegan_eval.py:62-103 (one pass) and
denoise_eavl_iter.py:62-114 (output fed back `num_iterations` times, intermediates saved):

    for every *.png/*.jpg/*.jpeg in input_dir:
        open -> RGB -> bicubic resize to image_size -> ToTensor -> Normalize(0.5, 0.5)      eval.py:91-92
        generator(x)                                                                        eval.py:94-95
        y*0.5+0.5 -> ToPILImage (mul(255).byte(): truncation) -> save under the same name   eval.py:97-99

The reference pushes one image at a time through the network; here the decoded images are grouped into batches and go
through `HostPipeline` (upload, forward and download overlapped; uint8 both ways, normalisation and the uint8 view fused
into the first/last kernel).  Decoding, resizing and encoding stay on the CPU with PIL exactly as in the reference (they
are image I/O, not the hot path); a small thread pool keeps them from starving the GPU.
"""
from __future__ import annotations

import os
from concurrent.futures import ThreadPoolExecutor
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

EXTENSIONS = (".png", ".jpg", ".jpeg")


def _load_rgb(path: str, image_size: Tuple[int, int]) -> np.ndarray:
    from PIL import Image

    with Image.open(path) as im:
        return np.asarray(im.convert("RGB").resize(image_size, resample=Image.Resampling.BICUBIC), dtype=np.uint8)


def _save_rgb(arr: np.ndarray, path: str) -> None:
    from PIL import Image

    Image.fromarray(arr, mode="RGB").save(path)


def enhance_images(checkpoint_path, input_dir: str = "testNoise", output_dir: str = "testOp",
                   image_size: Tuple[int, int] = (256, 256), num_iterations: int = 1, batch_size: int = 64,
                   save_intermediates: Optional[bool] = None, model=None, workers: int = 8) -> List[str]:
    """Denoise every image of `input_dir` into `output_dir`; returns the list of files written.

    checkpoint_path: what the reference passes to torch.load (the trainer's {"generator": state_dict, ...} file, eval.py:68-69),
    or None with `model=` an already loaded DenoiseGenerator.  num_iterations == 1 saves `<name>` like denoisegan_eval.py;
    num_iterations > 1 follows denoise_eavl_iter.py: `<base>_iter<i><ext>` per iteration (unless save_intermediates=False) and
    `<base>_final<ext>`.  An image that cannot be read is reported and skipped, like the reference's try/except."""
    from . import api
    from .pipeline import HostPipeline

    if num_iterations < 1:
        raise ValueError("num_iterations must be >= 1")
    if model is None:
        model = api.load(checkpoint_path, strict=True)
    if save_intermediates is None:
        save_intermediates = num_iterations > 1
    os.makedirs(output_dir, exist_ok=True)
    names = [f for f in os.listdir(input_dir) if f.lower().endswith(EXTENSIONS)]
    written: List[str] = []
    dev = next(model.parameters()).device
    pipe = HostPipeline(model)
    with ThreadPoolExecutor(max_workers=max(1, workers)) as pool:
        for b0 in range(0, len(names), batch_size):
            chunk = names[b0:b0 + batch_size]
            decoded = list(pool.map(lambda f: _try(_load_rgb, os.path.join(input_dir, f), image_size), chunk))
            ok = [(f, a) for f, a in zip(chunk, decoded) if not isinstance(a, Exception)]
            for f, a in zip(chunk, decoded):
                if isinstance(a, Exception):
                    print(f"Error processing image {os.path.join(input_dir, f)}: {a}")
            if not ok:
                continue
            batch = np.stack([a for _, a in ok])
            jobs = []
            if num_iterations == 1:
                out = next(iter(pipe.run([batch])))
                jobs = [(out[k].numpy(), os.path.join(output_dir, f)) for k, (f, _) in enumerate(ok)]
            else:
                # iterate on the device; every iteration's uint8 view comes back for the intermediate files
                z = model.forward_u8(torch.from_numpy(batch).to(dev), out_u8=False)
                for it in range(1, num_iterations + 1):
                    if it > 1:
                        z = model(z)
                    last = it == num_iterations
                    if save_intermediates or last:
                        view = model.view_u8(z).cpu().numpy()     # the reference's *0.5+0.5 / clamp / ToPILImage view, one HIP kernel (cid_view_u8)
                        for k, (f, _) in enumerate(ok):
                            base, ext = os.path.splitext(f)
                            if save_intermediates:
                                jobs.append((view[k], os.path.join(output_dir, f"{base}_iter{it}{ext}")))
                            if last:
                                jobs.append((view[k], os.path.join(output_dir, f"{base}_final{ext}")))
            for res, (_, path) in zip(pool.map(lambda j: _try(_save_rgb, j[0], j[1]), jobs), jobs):
                if isinstance(res, Exception):
                    print(f"Error saving image {path}: {res}")
                else:
                    written.append(path)
    return written


def _try(fn, *args):
    try:
        return fn(*args)
    except Exception as e:   # noqa: BLE001 - reported per image by the caller, like the reference's loop
        return e
