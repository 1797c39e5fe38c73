"""Ad-hoc: compare the z planes the fused upconv1[0] kernel leaves in the arena with a CPU computation."""
import os, sys, ctypes
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../../..")
import celebrity_image_denoiser_amd as cid
from celebrity_image_denoiser_amd import synth, _lib
from oracle import torch_oracle
wset = "hot"
sd = synth.make_state_dict(wset)
for (n, h, w) in ((1, 16, 16), (2, 128, 128)):
    x, _, _ = synth.make_batch(n, h, w, first_index=3)
    m = cid.load(sd, device="cuda:0", strict=True)
    y = m(torch.from_numpy(x).to("cuda:0")); torch.cuda.synchronize()
    out, st = torch_oracle.forward(sd, x, return_stages=True)
    e1 = st["down1"][:, :, : st["up1"].shape[2], : st["up1"].shape[3]]
    cat1 = torch.cat([st["up1"], e1], dim=1)
    t4 = F.relu(F.conv2d(cat1, torch.from_numpy(sd["upconv1.0.weight"]), torch.from_numpy(sd["upconv1.0.bias"]), padding=1))
    W2 = torch.from_numpy(sd["upconv1.2.weight"])          # [3,64,3,3]
    zexp = torch.einsum("nchw,octx->ntxohw", t4, W2).reshape(n, 27, t4.shape[2], t4.shape[3])   # plane = (3*ty+tx)*3+co
    L = _lib.lib()
    off, c, hs, ws, ps, coff = ctypes.c_size_t(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    L.cid_stage_view(b"upconv1.0", n, h, w, ctypes.byref(off), ctypes.byref(c), ctypes.byref(hs), ctypes.byref(ws), ctypes.byref(ps), ctypes.byref(coff))
    z = m._ws[off.value:].view(torch.float32)[: n * 27 * hs.value * ws.value].view(n, 27, hs.value, ws.value).cpu()
    print((n, h, w), "z max|got|", float(z.abs().max()), "max|exp|", float(zexp.abs().max()), "max|diff|", float((z - zexp).abs().max()))
    bad = (z - zexp).abs().amax(dim=(0, 2, 3))
    print(" per-plane max diff:", [round(float(v), 5) for v in bad])
    print(" out diff", float((y.cpu() - out).abs().max()))
    rows = (z - zexp).abs().amax(dim=(0, 1, 3)); cols = (z - zexp).abs().amax(dim=(0, 1, 2))
    print(" bad rows:", [i for i, v in enumerate(rows) if v > 1e-4][:40]); print(" bad cols:", [i for i, v in enumerate(cols) if v > 1e-4][:40])
