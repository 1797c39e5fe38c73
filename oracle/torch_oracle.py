"""TEST INFRASTRUCTURE — CPU oracle #1 for the denoise forward (ATen / oneDNN arithmetic).

This is a functional restatement of the reference's `DenoiseGenerator.forward`
(reference backend/app.py:80-103) on top of the same third-party ATen CPU operators the
reference's nn.Module dispatches to (`conv2d`, `relu`, `max_pool2d`, `conv_transpose2d`, `cat`,
`tanh`; reference backend/app.py:42-78).  It is written from the op list, not copied: it has no
nn.Module, takes a plain state_dict and optionally returns every intermediate the golden
fixtures record.

Parity pinned: tests/test_oracle.py checks it bit-for-bit against tests/golden/*.npz, which
tests/golden/make_golden.py generated in the build container by lifting the reference class
itself out of /root/reference/backend/app.py at run time.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The shipped package never does.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

# names of the recorded intermediates = the reference module attributes they come out of
STAGES = ("down1", "pool1", "down2", "pool2", "bottleneck", "up2", "upconv2", "up1", "upconv1", "out")


def _t(sd, key, dtype):
    v = sd[key]
    if isinstance(v, np.ndarray):
        v = torch.from_numpy(np.ascontiguousarray(v))
    return v.detach().to(device="cpu", dtype=dtype)


def forward(state_dict, x, return_stages: bool = False, dtype=torch.float32):
    """x: [N,3,H,W] (numpy or torch, any float dtype) -> [N,3,4*(H//4),4*(W//4)] torch CPU tensor.

    dtype=torch.float64 gives the "exact" value used to rank fp32 implementations by error.
    """
    if isinstance(x, np.ndarray):
        x = torch.from_numpy(np.ascontiguousarray(x))
    x = x.detach().to(device="cpu", dtype=dtype)
    p = lambda k: _t(state_dict, k, dtype)  # noqa: E731
    st = OrderedDict()
    with torch.no_grad():
        def block(t, name):  # Conv-ReLU-Conv-ReLU   (app.py:42-47, 50-55, 58-63, 66-71)
            t = F.relu(F.conv2d(t, p(name + ".0.weight"), p(name + ".0.bias"), padding=1))
            return F.relu(F.conv2d(t, p(name + ".2.weight"), p(name + ".2.bias"), padding=1))

        e1 = block(x, "down1")                                   # app.py:81
        p1 = F.max_pool2d(e1, 2, 2)                              # app.py:82
        e2 = block(p1, "down2")                                  # app.py:84
        p2 = F.max_pool2d(e2, 2, 2)                              # app.py:85
        b = block(p2, "bottleneck")                              # app.py:87
        d2 = F.conv_transpose2d(b, p("up2.weight"), p("up2.bias"), stride=2)   # app.py:89
        st.update(down1=e1, pool1=p1, down2=e2, pool2=p2, bottleneck=b, up2=d2)
        if d2.shape != e2.shape:                                 # app.py:90-92 top-left crop
            e2 = e2[:, :, : d2.shape[2], : d2.shape[3]]
        d2 = block(torch.cat([d2, e2], dim=1), "upconv2")        # app.py:93-94  [up, skip]
        d1 = F.conv_transpose2d(d2, p("up1.weight"), p("up1.bias"), stride=2)  # app.py:96
        st.update(upconv2=d2, up1=d1)
        if d1.shape != e1.shape:                                 # app.py:97-99
            e1 = e1[:, :, : d1.shape[2], : d1.shape[3]]
        d1 = torch.cat([d1, e1], dim=1)                          # app.py:100
        d1 = F.relu(F.conv2d(d1, p("upconv1.0.weight"), p("upconv1.0.bias"), padding=1))
        d1 = F.conv2d(d1, p("upconv1.2.weight"), p("upconv1.2.bias"), padding=1)  # app.py:75-77,101
        out = torch.tanh(d1)                                     # app.py:103
        st.update(upconv1=d1, out=out)
    return (out, st) if return_stages else out


def psnr(a, b, data_range: float = 2.0) -> float:
    """mean over the batch of 10*log10(data_range^2 / MSE_i): what
    skimage.metrics.peak_signal_noise_ratio(.., data_range=2.0) computes per image in
    reference backend/trainingcode/denoise_gan_code/training.py:378-383 (float64 arithmetic)."""
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    mse = ((a - b) ** 2).flatten(1).mean(dim=1)
    return float((10.0 * torch.log10((data_range ** 2) / mse)).mean())
