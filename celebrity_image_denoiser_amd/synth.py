"""Portable synthetic weights and inputs for the denoise hot path.

The reference ships no trained denoise checkpoint in this tree and its server falls back to
random-init weights when the checkpoint is missing (reference backend/app.py:327-336), so every
parity test and the benchmark run on *seeded synthetic* weights.  Torch's RNG streams are not
portable between builds, so everything here comes from a counter-based splitmix64 hash: the same
(seed, key, index) gives the same float on every machine, with numpy only.

Inputs follow the reference's data recipe:
  * noisy = clip(clean + N(0, sigma=25), 0, 255).astype(uint8)
        reference backend/trainingcode/denoise_gan_code/noise_generation.py:6-10
  * x = (noisy/255 - 0.5)/0.5 as float32 NCHW
        reference backend/trainingcode/denoise_gan_code/training.py:152-155, backend/app.py:401-405
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)

WEIGHT_SEED = 0x5EED
CLEAN_SEED = 0xC0FFEE
NOISE_SEED = 0xBADD1E

# (key prefix, kind, Cin, Cout) in reference declaration order, backend/app.py:42-78.
LAYERS = (
    ("down1.0", "conv", 3, 64),
    ("down1.2", "conv", 64, 64),
    ("down2.0", "conv", 64, 128),
    ("down2.2", "conv", 128, 128),
    ("bottleneck.0", "conv", 128, 256),
    ("bottleneck.2", "conv", 256, 256),
    ("up2", "convT", 256, 128),
    ("upconv2.0", "conv", 256, 128),
    ("upconv2.2", "conv", 128, 128),
    ("up1", "convT", 128, 64),
    ("upconv1.0", "conv", 128, 64),
    ("upconv1.2", "conv", 64, 3),
)


def param_shapes() -> "OrderedDict[str, tuple]":
    """state_dict key -> shape, same names/shapes as the reference module (app.py:39-78)."""
    out: "OrderedDict[str, tuple]" = OrderedDict()
    for name, kind, cin, cout in LAYERS:
        if kind == "conv":
            out[name + ".weight"] = (cout, cin, 3, 3)  # nn.Conv2d: [Cout, Cin, kh, kw]
        else:
            out[name + ".weight"] = (cin, cout, 2, 2)  # nn.ConvTranspose2d: [Cin, Cout, kh, kw]
        out[name + ".bias"] = (cout,)
    return out


def splitmix64(x: np.ndarray) -> np.ndarray:
    """Vectorised splitmix64 finaliser on uint64 arrays (wrapping arithmetic)."""
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def _fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for b in s.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def hash_uniform(seed: int, stream: int, n: int, offset: int = 0) -> np.ndarray:
    """n doubles in [0,1): element i = top 53 bits of splitmix64(splitmix64(seed^stream) + offset+i)."""
    base = splitmix64(np.array([(seed ^ stream) & 0xFFFFFFFFFFFFFFFF], dtype=np.uint64))[0]
    idx = np.arange(offset, offset + n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = splitmix64(base + idx)
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def make_state_dict(kind: str = "default", seed: int = WEIGHT_SEED) -> "OrderedDict[str, np.ndarray]":
    """Synthetic float32 weights keyed like the reference state_dict.

    kind="default": U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weights and biases — the scale of
        PyTorch's default Conv2d init, i.e. of the server's random-init fallback (app.py:333-336).
    kind="hot": He-uniform weights U(+-sqrt(6/fan_in)), biases U(+-0.05): keeps activation
        variance through the ReLU stack so ReLU sparsity is ~50 % and tanh leaves its linear range.
    """
    if kind not in ("default", "hot"):
        raise ValueError(f"unknown weight set {kind!r}")
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for key, shape in param_shapes().items():
        n = int(np.prod(shape))
        layer = key.rsplit(".", 1)[0]
        wshape = param_shapes()[layer + ".weight"]
        fan_in = wshape[1] * wshape[2] * wshape[3]  # torch's rule, also for ConvTranspose2d
        if kind == "default":
            bound = 1.0 / math.sqrt(fan_in)
        elif key.endswith(".weight"):
            bound = math.sqrt(6.0 / fan_in)
        else:
            bound = 0.05
        u = hash_uniform(seed, _fnv1a64(kind + ":" + key), n)
        sd[key] = ((2.0 * u - 1.0) * bound).astype(np.float32).reshape(shape)
    return sd


def clean_images_u8(n: int, h: int, w: int, first_index: int = 0) -> np.ndarray:
    """[n,h,w,3] uint8 smooth "face-like" fields: an 8x8x3 hash grid, bilinearly upsampled."""
    out = np.empty((n, h, w, 3), dtype=np.uint8)
    ys = np.linspace(0.0, 7.0, h) if h > 1 else np.zeros(1)
    xs = np.linspace(0.0, 7.0, w) if w > 1 else np.zeros(1)
    y0 = np.minimum(ys.astype(np.int64), 6)
    x0 = np.minimum(xs.astype(np.int64), 6)
    fy = (ys - y0)[:, None, None]
    fx = (xs - x0)[None, :, None]
    for i in range(n):
        g = np.floor(hash_uniform(CLEAN_SEED + first_index + i, 0, 8 * 8 * 3) * 256.0).reshape(8, 8, 3)
        a = g[y0][:, x0]
        b = g[y0][:, x0 + 1]
        c = g[y0 + 1][:, x0]
        d = g[y0 + 1][:, x0 + 1]
        v = (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy
        out[i] = np.clip(np.floor(v + 0.5), 0, 255).astype(np.uint8)
    return out


def add_gaussian_noise(clean_u8: np.ndarray, sigma: float = 25.0, first_index: int = 0) -> np.ndarray:
    """clip(img + N(0,sigma), 0, 255).astype(uint8) per image (noise_generation.py:6-10), with the
    normal deviates from Box-Muller on the hash stream so the result is machine-independent."""
    n = clean_u8.shape[0]
    per = int(np.prod(clean_u8.shape[1:]))
    out = np.empty_like(clean_u8)
    for i in range(n):
        u1 = hash_uniform(NOISE_SEED + first_index + i, 1, per)
        u2 = hash_uniform(NOISE_SEED + first_index + i, 2, per)
        z = np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)
        noisy = clean_u8[i].astype(np.float64).reshape(-1) + sigma * z
        out[i] = np.clip(noisy, 0, 255).astype(np.uint8).reshape(clean_u8.shape[1:])
    return out


def normalize_u8(img_u8_nhwc: np.ndarray) -> np.ndarray:
    """uint8 NHWC -> float32 NCHW in [-1,1]: ToTensor (/255) then Normalize(0.5,0.5)
    (app.py:401-405; training.py:152-155), both steps in float32 like torchvision."""
    t = img_u8_nhwc.astype(np.float32) / np.float32(255.0)
    t = (t - np.float32(0.5)) / np.float32(0.5)
    return np.ascontiguousarray(t.transpose(0, 3, 1, 2))


def make_batch(n: int, h: int, w: int, first_index: int = 0, sigma: float = 25.0):
    """(x_noisy f32 NCHW, clean f32 NCHW, noisy u8 NHWC) for images first_index..first_index+n-1."""
    clean = clean_images_u8(n, h, w, first_index)
    noisy = add_gaussian_noise(clean, sigma, first_index)
    return normalize_u8(noisy), normalize_u8(clean), noisy
