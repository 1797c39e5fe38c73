"""Ad-hoc (CPU): would Winograd F(2x2,3x3) at fp16 storage keep the 5e-3 bound?  Emulates the whole network with every rounding to half
the fp16 kernels make (direct: reproduces the measured error) and with the 3x3 layers as Winograd on half V and U.  Result (hot weights):
direct 2.4e-3, Winograd 3.1e-3 (3.3e-3 with half-arithmetic transforms).  See DESIGN.md section 7 for why it is not built."""
import sys, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, '/root/repo')
from celebrity_image_denoiser_amd import synth
from oracle import torch_oracle
torch.set_num_threads(8)
h16 = lambda t: t.half().float()
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)

def conv_direct(x, w, b):           # x already half-rounded values in fp32
    return F.conv2d(x, h16(w), b, padding=1)

def conv_wino(x, w, b, v_half=True):
    N, C, H, W = x.shape
    K = w.shape[0]
    U = torch.einsum('ai,kcij,bj->abkc', G, w.double(), G).float()
    U = h16(U)
    Hp, Wp = (H + 1) // 2 * 2, (W + 1) // 2 * 2
    xp = F.pad(x, (1, 1 + Wp - W, 1, 1 + Hp - H))
    # tiles: 4x4 patches stride 2
    pt = xp.unfold(2, 4, 2).unfold(3, 4, 2)             # N,C,th,tw,4,4
    V = torch.einsum('ai,nctuij,bj->abnctu', Bt, pt, Bt)
    if v_half: V = h16(V)
    M = torch.einsum('abnctu,abkc->abnktu', V, U)       # fp32 accumulate
    Y = torch.einsum('ia,abnktu,jb->nktiuj', At, M, At)  # N,K,th,2,tw,2
    Y = Y.reshape(N, K, Hp, Wp)[:, :, :H, :W]
    return Y + b.view(1, -1, 1, 1)

def forward(sd, x, mode):
    p = lambda k: torch.from_numpy(sd[k])
    c3 = {"direct": conv_direct, "wino": conv_wino}[mode]
    def blk(t, name, first=False):
        if first:
            t = F.relu(F.conv2d(t, p(name + ".0.weight"), p(name + ".0.bias"), padding=1))   # head: fp32 arithmetic, half store
        else:
            t = F.relu(c3(t, p(name + ".0.weight"), p(name + ".0.bias")))
        t = h16(t)
        t = F.relu(c3(t, p(name + ".2.weight"), p(name + ".2.bias")))
        return h16(t)
    e1 = blk(x, "down1", True); p1 = F.max_pool2d(e1, 2)
    e2 = blk(p1, "down2"); p2 = F.max_pool2d(e2, 2)
    b = blk(p2, "bottleneck")
    d2 = h16(F.conv_transpose2d(b, h16(p("up2.weight")), p("up2.bias"), stride=2))
    d2 = blk(torch.cat([d2, e2], 1), "upconv2")
    d1 = h16(F.conv_transpose2d(d2, h16(p("up1.weight")), p("up1.bias"), stride=2))
    t = h16(F.relu(c3(torch.cat([d1, e1], 1), p("upconv1.0.weight"), p("upconv1.0.bias"))))
    out = F.conv2d(t, h16(p("upconv1.2.weight")), p("upconv1.2.bias"), padding=1)      # tail: fp16 MFMA on half input
    return torch.tanh(out)

for wset in ("default", "hot"):
    sd = synth.make_state_dict(wset)
    x, _, _ = synth.make_batch(2, 128, 128, 100)
    ref = torch_oracle.forward(sd, x)
    xt = torch.from_numpy(x)
    with torch.no_grad():
        yd = forward(sd, xt, "direct"); yw = forward(sd, xt, "wino")
    print(wset, "direct-fp16 max|d| %.3e   winograd-fp16 max|d| %.3e   (wino vs direct %.3e)" % ((yd - ref).abs().max(), (yw - ref).abs().max(), (yw - yd).abs().max()))
