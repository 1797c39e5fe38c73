"""Ad-hoc (CPU): what does rounding the last layer's 27 per-tap partial sums z to half cost the fp16-storage path?  Emulates the whole
network with every rounding to half the fp16 kernels make (as emulate_f16_winograd.py), then the last layer three ways: unfused
(k_conv_tail_h: fp32 accumulation of all 576 products), z per tap in fp32 (the fp32 path's fused form) and z per tap rounded to half
(k_conv3x3_h16<128,64,ZOUT> + k_conv_tail_zh, round 4).  Stated tolerance of the path: max|delta| <= 5e-3 on He-gain weights."""
import sys, os, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
from celebrity_image_denoiser_amd import synth
from oracle import torch_oracle
torch.set_num_threads(8)
h16 = lambda t: t.half().float()


def trunk(sd, x):
    p = lambda k: torch.from_numpy(sd[k])
    c3 = lambda t, w, b: F.conv2d(t, h16(w), b, padding=1)
    def blk(t, name, first=False):
        t = F.relu(F.conv2d(t, p(name + ".0.weight"), p(name + ".0.bias"), padding=1)) if first else F.relu(c3(t, p(name + ".0.weight"), p(name + ".0.bias")))
        return h16(F.relu(c3(h16(t), p(name + ".2.weight"), p(name + ".2.bias"))))
    e1 = blk(x, "down1", True); p1 = F.max_pool2d(e1, 2)
    e2 = blk(p1, "down2"); p2 = F.max_pool2d(e2, 2)
    b = blk(p2, "bottleneck")
    d2 = h16(F.conv_transpose2d(b, h16(p("up2.weight")), p("up2.bias"), stride=2))
    d2 = blk(torch.cat([d2, e2], 1), "upconv2")
    d1 = h16(F.conv_transpose2d(d2, h16(p("up1.weight")), p("up1.bias"), stride=2))
    return h16(F.relu(c3(torch.cat([d1, e1], 1), p("upconv1.0.weight"), p("upconv1.0.bias"))))


for wset in ("default", "hot"):
    sd = synth.make_state_dict(wset)
    x, _, _ = synth.make_batch(16, 128, 128, 100)
    ref = torch_oracle.forward(sd, x)
    with torch.no_grad():
        t = trunk(sd, torch.from_numpy(x))
        w2, b2 = h16(torch.from_numpy(sd["upconv1.2.weight"])), torch.from_numpy(sd["upconv1.2.bias"])
        unfused = torch.tanh(F.conv2d(t, w2, b2, padding=1))
        # z[tap] = 1x1 contraction with tap's [3 x 64] weights; out = sum of the nine shifted z
        outs = {}
        for name, rnd in (("z fp32", lambda v: v), ("z half", h16)):
            acc = b2.view(1, 3, 1, 1).expand(t.shape[0], 3, t.shape[2], t.shape[3]).clone()
            for kh in range(3):
                for kw in range(3):
                    z = rnd(F.conv2d(t, w2[:, :, kh:kh + 1, kw:kw + 1]))
                    acc += F.pad(z, (1, 1, 1, 1))[:, :, kh:kh + t.shape[2], kw:kw + t.shape[3]]
            outs[name] = torch.tanh(acc)
        zmax = max(float(F.conv2d(t, w2[:, :, kh:kh + 1, kw:kw + 1]).abs().max()) for kh in range(3) for kw in range(3))
    print(wset, "vs fp32 oracle: unfused %.3e   z fp32 %.3e   z half %.3e   | z-half vs unfused %.3e   max|z| %.2f" % (
        (unfused - ref).abs().max(), (outs["z fp32"] - ref).abs().max(), (outs["z half"] - ref).abs().max(), (outs["z half"] - unfused).abs().max(), zmax))
