"""PSNR as the reference's denoise trainer defines it.

reference backend/trainingcode/denoise_gan_code/training.py:378-383:
    psnr(denoised_np[i], clean_np[i], data_range=2.0), mean over the batch,
with skimage's peak_signal_noise_ratio = 10*log10(data_range^2 / mse) in float64.
"""
from __future__ import annotations

import numpy as np


def psnr(a, b, data_range: float = 2.0) -> float:
    """Mean over the batch dimension of 10*log10(data_range**2 / MSE_i); inputs [-1,1] tensors/arrays."""
    a = _np(a).astype(np.float64)
    b = _np(b).astype(np.float64)
    if a.shape != b.shape:
        raise ValueError("Input images must have the same dimensions.")
    mse = ((a - b) ** 2).reshape(a.shape[0], -1).mean(axis=1)
    with np.errstate(divide="ignore"):
        return float(np.mean(10.0 * np.log10((data_range ** 2) / mse)))


def _np(x):
    if isinstance(x, np.ndarray):
        return x
    return x.detach().cpu().numpy()
