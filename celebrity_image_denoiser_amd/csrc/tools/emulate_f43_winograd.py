"""Ad-hoc (CPU): would Winograd F(4x4,3x3) in fp32 keep the 1e-5 parity bound?  The eight 3x3 GEMM layers emulated in fp32 with the
transform matrices of Lavin & Gray (interpolation points 0, +-1, +-2, inf), U = G g G^T in double rounded once to fp32, V, M and the output
transform in fp32; compared with the float64 forward and with the fp32 oracle, final output and every stage."""
import sys, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..', '..', '..'))
from celebrity_image_denoiser_amd import synth
torch.set_num_threads(8)
G = torch.tensor([[1/4, 0, 0], [-1/6, -1/6, -1/6], [-1/6, 1/6, -1/6], [1/24, 1/12, 1/6], [1/24, -1/12, 1/6], [0, 0, 1]], dtype=torch.float64)
Bt = torch.tensor([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=torch.float64)
At = torch.tensor([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=torch.float64)
G2 = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
Bt2 = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
At2 = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)

def conv_wino(x, w, b, dt, m):
    G_, Bt_, At_ = (G, Bt, At) if m == 4 else (G2, Bt2, At2)
    a = m + 2
    N, C, H, W = x.shape
    K = w.shape[0]
    U = torch.einsum('ai,kcij,bj->abkc', G_, w.double(), G_).to(dt)
    Hp, Wp = (H + m - 1) // m * m, (W + m - 1) // m * m
    xp = F.pad(x, (1, 1 + Wp - W, 1, 1 + Hp - H))
    pt = xp.unfold(2, a, m).unfold(3, a, m)                       # N,C,th,tw,a,a
    V = torch.einsum('ai,nctuij,bj->abnctu', Bt_.to(dt), pt, Bt_.to(dt))
    M = torch.einsum('abnctu,abkc->abnktu', V, U)
    Y = torch.einsum('ia,abnktu,jb->nktiuj', At_.to(dt), M, At_.to(dt))
    Y = Y.reshape(N, K, Hp, Wp)[:, :, :H, :W]
    return Y + b.view(1, -1, 1, 1).to(dt)

def forward(sd, x, dt, mode):
    p = lambda k: torch.from_numpy(sd[k]).to(dt)
    c3 = (lambda t, w, b: F.conv2d(t, w, b, padding=1)) if mode == 0 else (lambda t, w, b: conv_wino(t, w, b, dt, mode))
    st = {}
    def blk(t, name, first=False):
        t = F.relu((F.conv2d(t, p(name + ".0.weight"), p(name + ".0.bias"), padding=1)) if first else c3(t, p(name + ".0.weight"), p(name + ".0.bias")))
        return F.relu(c3(t, p(name + ".2.weight"), p(name + ".2.bias")))
    e1 = blk(x, "down1", True); st["down1"] = e1; p1 = F.max_pool2d(e1, 2)
    e2 = blk(p1, "down2"); st["down2"] = e2; p2 = F.max_pool2d(e2, 2)
    b = blk(p2, "bottleneck"); st["bottleneck"] = b
    d2 = F.conv_transpose2d(b, p("up2.weight"), p("up2.bias"), stride=2)
    d2 = blk(torch.cat([d2, e2], 1), "upconv2"); st["upconv2"] = d2
    d1 = F.conv_transpose2d(d2, p("up1.weight"), p("up1.bias"), stride=2)
    t = F.relu(c3(torch.cat([d1, e1], 1), p("upconv1.0.weight"), p("upconv1.0.bias"))); st["upconv1.0"] = t
    out = torch.tanh(F.conv2d(t, p("upconv1.2.weight"), p("upconv1.2.bias"), padding=1)); st["out"] = out
    return st

for wset in ("default", "hot"):
    sd = synth.make_state_dict(wset)
    x, _, _ = synth.make_batch(2, 128, 128, 100)
    with torch.no_grad():
        r64 = forward(sd, torch.from_numpy(x).double(), torch.float64, 0)
        d32 = forward(sd, torch.from_numpy(x), torch.float32, 0)
        w2 = forward(sd, torch.from_numpy(x), torch.float32, 2)
        w4 = forward(sd, torch.from_numpy(x), torch.float32, 4)
    print(wset)
    for k in r64:
        s = max(1.0, r64[k].abs().max().item())
        print("  %-11s |stage| %.2f   direct-fp32 %.2e   F(2x2) %.2e   F(4x4) %.2e   (bound 1e-5*max(1,|stage|) = %.1e)" % (
            k, r64[k].abs().max(), (d32[k] - r64[k]).abs().max(), (w2[k] - r64[k]).abs().max(), (w4[k] - r64[k]).abs().max(), 1e-5 * s))

# iterated use (denoise_eavl_iter.py:93-96): the output is fed back three times
print("iterated, hot weights, 2x3x64x64")
sd = synth.make_state_dict("hot")
x, _, _ = synth.make_batch(2, 64, 64, 7)
with torch.no_grad():
    a64, a32, a2, a4 = torch.from_numpy(x).double(), torch.from_numpy(x), torch.from_numpy(x), torch.from_numpy(x)
    for it in range(3):
        a64 = forward(sd, a64, torch.float64, 0)["out"]; a32 = forward(sd, a32, torch.float32, 0)["out"]
        a2 = forward(sd, a2, torch.float32, 2)["out"]; a4 = forward(sd, a4, torch.float32, 4)["out"]
        print("  iteration %d: direct-fp32 %.2e   F(2x2) %.2e   F(4x4) %.2e   F(4x4) vs direct-fp32 %.2e" % (
            it + 1, (a32 - a64).abs().max(), (a2 - a64).abs().max(), (a4 - a64).abs().max(), (a4 - a32).abs().max()))
