"""TEST INFRASTRUCTURE — ctypes wrapper around oracle/libdenoise_oracle.so (oracle/denoise_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from collections import OrderedDict

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libdenoise_oracle.so")
STAGES = ("down1", "pool1", "down2", "pool2", "bottleneck", "up2", "upconv2", "up1", "upconv1")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "denoise_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "all"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        L.cid_oracle_param_count.restype = ctypes.c_size_t
        L.cid_oracle_forward.restype = ctypes.c_int
        L.cid_oracle_forward.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        _lib = L
    return _lib


def flatten_params(state_dict) -> np.ndarray:
    """state_dict (numpy or torch values) -> flat fp32 blob in the reference's key order."""
    keys = []
    for name in ("down1.0", "down1.2", "down2.0", "down2.2", "bottleneck.0", "bottleneck.2", "up2",
                 "upconv2.0", "upconv2.2", "up1", "upconv1.0", "upconv1.2"):
        keys += [name + ".weight", name + ".bias"]
    parts = []
    for k in keys:
        v = state_dict[k]
        if not isinstance(v, np.ndarray):
            v = v.detach().cpu().numpy()
        parts.append(np.ascontiguousarray(v, dtype=np.float32).reshape(-1))
    blob = np.concatenate(parts)
    assert blob.size == lib().cid_oracle_param_count()
    return blob


def forward(state_dict, x, acc64: bool = False, return_stages: bool = False):
    """x [N,3,H,W] float32 numpy -> out [N,3,4*(H//4),4*(W//4)] float32 numpy (+ OrderedDict of stages)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    n, c, h, w = x.shape
    if c != 3:
        raise ValueError("expected [N,3,H,W]")
    if h < 4 or w < 4:
        raise ValueError("H and W must be >= 4")
    blob = flatten_params(state_dict)
    h1, w1 = h // 2, w // 2
    h2, w2 = h1 // 2, w1 // 2
    out = np.empty((n, 3, 4 * h2, 4 * w2), dtype=np.float32)
    shapes = [(n, 64, h, w), (n, 64, h1, w1), (n, 128, h1, w1), (n, 128, h2, w2), (n, 256, h2, w2),
              (n, 128, 2 * h2, 2 * w2), (n, 128, 2 * h2, 2 * w2), (n, 64, 4 * h2, 4 * w2), (n, 3, 4 * h2, 4 * w2)]
    st_arrays, st_ptr = None, None
    if return_stages:
        st_arrays = [np.empty(s, dtype=np.float32) for s in shapes]
        st_ptr = (ctypes.c_void_p * 9)(*[a.ctypes.data for a in st_arrays])
    rc = lib().cid_oracle_forward(blob.ctypes.data, x.ctypes.data, out.ctypes.data, n, h, w, int(acc64), st_ptr)
    if rc != 0:
        raise RuntimeError(f"cid_oracle_forward failed with status {rc}")
    if return_stages:
        st = OrderedDict(zip(STAGES, st_arrays))
        st["out"] = out
        return out, st
    return out
