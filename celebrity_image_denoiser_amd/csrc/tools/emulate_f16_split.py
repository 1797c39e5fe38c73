"""Ad-hoc (CPU): the numerics of a SPLIT-OPERAND form of the fp32 path — every fp32 operand x = hi + lo with hi = half(x), lo = half(x - hi)
(weights pre-scaled per layer by a power of two so that lo stays a normal half), every product as three fp16 MFMA products
hi*hi + hi*lo + lo*hi with fp32 accumulation (lo*lo, 2^-22 relative, dropped).  Direct convolution — no Winograd.  Compared against a
float64 evaluation of the same network, beside plain fp32 (ATen) and the error budget of the product's Winograd F(4x2) path
(4e-6 on white noise, profiles/r04_stress_checks.txt).  Stated tolerance of the fp32 path: max|delta| <= 1e-5.
This prices an IDEA; no kernel of the library computes this way."""
import sys, os, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
from celebrity_image_denoiser_amd import synth
torch.set_num_threads(8)


def split(t):
    hi = t.half().float()
    lo = (t - hi).half().float()
    return hi, lo


def scale_pow2(w):   # power of two that puts max|w| near 2^3: the low piece of a weight then stays far above the smallest normal half
    return 2.0 ** (3 - int(np.ceil(np.log2(float(w.abs().max())))))


def conv_split(x, w, b, transposed=False, drop_lolo=True):
    s = scale_pow2(w)
    xh, xl = split(x)
    wh, wl = split(w * s)
    op = (lambda a, k: F.conv_transpose2d(a, k, stride=2)) if transposed else (lambda a, k: F.conv2d(a, k, padding=1))
    y = op(xh, wh) + (op(xh, wl) + op(xl, wh))
    if not drop_lolo:
        y = y + op(xl, wl)
    return y / s + b.view(1, -1, 1, 1)


def forward(sd, x, conv, dtype=torch.float32):
    p = lambda k: torch.from_numpy(sd[k]).to(dtype)
    c = lambda t, n: conv(t, p(n + ".weight"), p(n + ".bias"))
    ct = lambda t, n: conv(t, p(n + ".weight"), p(n + ".bias"), True)
    blk = lambda t, n: F.relu(c(F.relu(c(t, n + ".0")), n + ".2"))
    e1 = blk(x, "down1"); e2 = blk(F.max_pool2d(e1, 2), "down2"); b = blk(F.max_pool2d(e2, 2), "bottleneck")
    d2 = blk(torch.cat([ct(b, "up2"), e2], 1), "upconv2")
    d1 = torch.cat([ct(d2, "up1"), e1], 1)
    return torch.tanh(c(F.relu(c(d1, "upconv1.0")), "upconv1.2"))


def plain(x, w, b, transposed=False):
    return (F.conv_transpose2d(x, w, b, stride=2) if transposed else F.conv2d(x, w, b, padding=1))


for wset in ("default", "hot"):
    sd = synth.make_state_dict(wset)
    for label, x in (("synthetic faces", synth.make_batch(4, 128, 128, 100)[0]),
                     ("white noise", np.random.default_rng(7).uniform(-1, 1, size=(4, 3, 128, 128)).astype(np.float32))):
        with torch.no_grad():
            xt = torch.from_numpy(x)
            ref = forward(sd, xt.double(), plain, torch.float64)
            f32 = forward(sd, xt, plain)
            sp3 = forward(sd, xt, conv_split)
            sp4 = forward(sd, xt, lambda *a: conv_split(*a, drop_lolo=False))
        e = lambda y: float((y.double() - ref).abs().max())
        print(f"{wset:8s} {label:16s} vs float64:  fp32 ATen {e(f32):.3e}   split, 3 products {e(sp3):.3e}   split, 4 products {e(sp4):.3e}")
