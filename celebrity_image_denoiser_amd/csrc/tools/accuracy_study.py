"""Ad-hoc (GPU box): how far is every 3x3 algorithm of the fp32 path from the EXACT result?  float64 evaluation of the network on the CPU (oracle, dtype=float64)
as the reference; against it: the ATen fp32 CPU forward (what the reference module computes) and the library's four algorithms — Winograd F(4x2) (default), Winograd
F(2x2), the direct 9-tap kernel (all three on the exact-fp32 MFMA) and the opt-in split-operand form on the fp16 MFMA (conv_algo="split16").  16 images per case,
both weight sets, face-like synthetic images and white noise.  Columns: max, 99.9th percentile, RMS of |y - y64| over all output elements, in units of 1e-6
(the path's contract: max|delta| <= 10 against the fp32 oracle)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../../..")
import celebrity_image_denoiser_amd as cid
from celebrity_image_denoiser_amd import synth
from oracle import torch_oracle

torch.set_num_threads(os.cpu_count() or 8)
def stats(y, ref):
    d = np.abs(y.astype(np.float64) - ref).ravel()
    return d.max() * 1e6, np.quantile(d, 0.999) * 1e6, np.sqrt((d * d).mean()) * 1e6

print("%-8s %-12s %-22s %10s %10s %10s" % ("weights", "input", "implementation", "max", "p99.9", "rms"))
for wset in ("default", "hot"):
    sd = synth.make_state_dict(wset)
    m = cid.load(sd, device="cuda:0", strict=True)
    g = torch.Generator(device="cpu"); g.manual_seed(2024)
    inputs = (("faces", synth.make_batch(16, 128, 128, first_index=7000)[0]),
              ("white noise", (torch.rand((16, 3, 128, 128), generator=g) * 2 - 1).numpy().astype(np.float32)))
    for iname, x in inputs:
        ref64 = torch_oracle.forward(sd, x, dtype=torch.float64).numpy()
        rows = [("ATen fp32 (CPU)", torch_oracle.forward(sd, x).numpy())]
        for algo in ("winograd42", "winograd64", "direct", "split16"):
            m.conv_algo = algo
            m.tail_algo = {"direct": "tiles"}.get(algo, "fused")
            rows.append((algo + (" (opt-in)" if algo == "split16" else " (default)" if algo == "winograd42" else ""), m(torch.from_numpy(x).to("cuda:0")).cpu().numpy()))
        for name, y in rows:
            print("%-8s %-12s %-22s %10.3f %10.3f %10.4f" % ((wset, iname, name) + stats(y, ref64)), flush=True)
