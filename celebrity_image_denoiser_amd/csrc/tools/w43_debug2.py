import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", ".."))
import celebrity_image_denoiser_amd as cid
from celebrity_image_denoiser_amd import synth
gd = os.path.join(os.path.dirname(__file__), "..", "..", "..", "tests", "golden")
m = cid.load(synth.make_state_dict("default"), device="cuda:0", strict=True)
m.conv_algo = "winograd43"; m.tail_algo = "fused"
g = np.load(os.path.join(gd, "tiny_default_16x16.npz"))
x = torch.from_numpy(g["x"]).to("cuda:0")
y = m(x)
n, _, h, w = g["x"].shape
got = m.stage_output("down1", n, h, w).cpu().numpy()
d = np.abs(got - g["down1"])
bad = np.argwhere(d > 1e-5)
print("bad count", len(bad), "of", d.size)
from collections import Counter
print("by (y,x):", sorted(Counter((int(b[2]), int(b[3])) for b in bad).items()))
print("by channel:", sorted(Counter(int(b[1]) for b in bad).items()))
print("by image:", Counter(int(b[0]) for b in bad))
