"""Ad-hoc: max|delta| of every 3x3 algorithm against the ATen oracle (fp32 and fp64) on WHITE-NOISE inputs — the worst case for Winograd's
error, away from the image-like synthetic batch the stated tolerance is defined on."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../../..")
import celebrity_image_denoiser_amd as cid
from celebrity_image_denoiser_amd import synth
from oracle import torch_oracle
for wset in ("default", "hot"):
    sd = synth.make_state_dict(wset)
    m = cid.load(sd, device="cuda:0", strict=True)
    g = torch.Generator(device="cpu"); g.manual_seed(99)
    x = (torch.rand((6, 3, 128, 134), generator=g) * 2 - 1).contiguous()
    ref64 = torch_oracle.forward(sd, x.numpy(), dtype=torch.float64).numpy() if "dtype" in torch_oracle.forward.__code__.co_varnames else None
    ref32 = torch_oracle.forward(sd, x.numpy()).numpy()
    for algo in ("winograd42", "winograd64", "direct", "split16"):
        m.conv_algo = algo; m.tail_algo = "tiles" if algo == "direct" else "fused"
        y = m(x.to("cuda:0")).cpu().numpy()
        print(wset, algo, "vs fp32 ATen", float(np.abs(y - ref32).max()), "vs fp64", None if ref64 is None else float(np.abs(y - ref64).max()), flush=True)
    if ref64 is not None: print(wset, "ATen fp32 vs fp64", float(np.abs(ref32 - ref64).max()))
