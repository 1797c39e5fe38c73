"""Ad-hoc (CPU): error of Winograd F(4x4,3x3) in fp32 on this network for several sets of interpolation points (Toom-Cook matrices
generated exactly with fractions), single pass and three fed-back iterations on the He-gain weights, against the float64 forward.
Result: profiles/r02_winograd_points.txt; the kernel uses 0, +-3/4, +-3/2, inf."""
import sys, numpy as np, torch, torch.nn.functional as F, itertools
from fractions import Fraction as Fr
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..', '..', '..'))
from celebrity_image_denoiser_amd import synth
torch.set_num_threads(8)

def polymul(a, b):
    r = [Fr(0)] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        for j, y in enumerate(b): r[i + j] += x * y
    return r

def matrices(points, m=4, r=3):
    n = m + r - 1
    pts = [Fr(p) for p in points]; assert len(pts) == n - 1
    AT = [[(pts[j] ** i if j < n - 1 else (Fr(1) if i == m - 1 else Fr(0))) for j in range(n)] for i in range(m)]
    G = []
    for j in range(n - 1):
        Fj = Fr(1)
        for l in range(n - 1):
            if l != j: Fj *= (pts[j] - pts[l])
        G.append([pts[j] ** k / Fj for k in range(r)])
    G.append([Fr(0)] * (r - 1) + [Fr(1)])
    BT = []
    for j in range(n - 1):
        poly = [Fr(1)]
        for l in range(n - 1):
            if l != j: poly = polymul(poly, [-pts[l], Fr(1)])
        BT.append(poly + [Fr(0)] * (n - len(poly)))
    poly = [Fr(1)]
    for l in range(n - 1): poly = polymul(poly, [-pts[l], Fr(1)])
    BT.append(poly)
    f = lambda M: torch.tensor([[float(x) for x in row] for row in M], dtype=torch.float64)
    return f(AT), f(G), f(BT)

def check(AT, G, BT):
    g = torch.randn(3, dtype=torch.float64); d = torch.randn(6, dtype=torch.float64)
    y = AT @ ((G @ g) * (BT @ d))
    ref = torch.stack([(d[i:i + 3] * g).sum() for i in range(4)])
    return (y - ref).abs().max().item()

def conv_wino(x, w, b, dt, mats):
    At_, G_, Bt_ = mats
    m, a = 4, 6
    N, C, H, W = x.shape; K = w.shape[0]
    U = torch.einsum('ai,kcij,bj->abkc', G_, w.double(), G_).to(dt)
    Hp, Wp = (H + m - 1) // m * m, (W + m - 1) // m * m
    xp = F.pad(x, (1, 1 + Wp - W, 1, 1 + Hp - H))
    pt = xp.unfold(2, a, m).unfold(3, a, m)
    V = torch.einsum('ai,nctuij,bj->abnctu', Bt_.to(dt), pt, Bt_.to(dt))
    M = torch.einsum('abnctu,abkc->abnktu', V, U)
    Y = torch.einsum('ia,abnktu,jb->nktiuj', At_.to(dt), M, At_.to(dt))
    return Y.reshape(N, K, Hp, Wp)[:, :, :H, :W] + b.view(1, -1, 1, 1).to(dt)

def forward(sd, x, dt, mats):
    p = lambda k: torch.from_numpy(sd[k]).to(dt)
    c3 = (lambda t, w, b: F.conv2d(t, w, b, padding=1)) if mats is None else (lambda t, w, b: conv_wino(t, w, b, dt, mats))
    def blk(t, name, first=False):
        t = F.relu((F.conv2d(t, p(name + ".0.weight"), p(name + ".0.bias"), padding=1)) if first else c3(t, p(name + ".0.weight"), p(name + ".0.bias")))
        return F.relu(c3(t, p(name + ".2.weight"), p(name + ".2.bias")))
    e1 = blk(x, "down1", True); p1 = F.max_pool2d(e1, 2)
    e2 = blk(p1, "down2"); p2 = F.max_pool2d(e2, 2)
    b = blk(p2, "bottleneck")
    d2 = F.conv_transpose2d(b, p("up2.weight"), p("up2.bias"), stride=2)
    d2 = blk(torch.cat([d2, e2], 1), "upconv2")
    d1 = F.conv_transpose2d(d2, p("up1.weight"), p("up1.bias"), stride=2)
    t = F.relu(c3(torch.cat([d1, e1], 1), p("upconv1.0.weight"), p("upconv1.0.bias")))
    return torch.tanh(F.conv2d(t, p("upconv1.2.weight"), p("upconv1.2.bias"), padding=1))

sd = synth.make_state_dict("hot")
x, _, _ = synth.make_batch(2, 64, 64, 7)
cands = {
  "0,1,-1,2,-2": [0, 1, -1, 2, -2],
  "0,1,-1,1/2,-1/2": [0, 1, -1, Fr(1, 2), Fr(-1, 2)],
  "0,1,-1,1/2,-2": [0, 1, -1, Fr(1, 2), -2],
  "0,1,-1,2,-1/2": [0, 1, -1, 2, Fr(-1, 2)],
  "0,1/2,-1/2,3/2,-3/2": [0, Fr(1, 2), Fr(-1, 2), Fr(3, 2), Fr(-3, 2)],
  "0,1,-1,3/2,-3/2": [0, 1, -1, Fr(3, 2), Fr(-3, 2)],
  "0,3/4,-3/4,3/2,-3/2": [0, Fr(3, 4), Fr(-3, 4), Fr(3, 2), Fr(-3, 2)],
  "0,1/2,-1/2,1,-2": [0, Fr(1, 2), Fr(-1, 2), 1, -2],
}
with torch.no_grad():
    r64 = []; a = torch.from_numpy(x).double()
    for it in range(3): a = forward(sd, a, torch.float64, None); r64.append(a)
    for name, pts in cands.items():
        mats = matrices(pts)
        ce = check(*mats)
        a = torch.from_numpy(x); errs = []
        for it in range(3):
            a = forward(sd, a, torch.float32, mats); errs.append((a - r64[it]).abs().max().item())
        print("%-22s identity err %.1e   iter errors %s" % (name, ce, " ".join("%.2e" % e for e in errs)), flush=True)
