#!/usr/bin/env python3
"""images/s of the forward over a sweep of batch sizes (one process, inputs resident in HBM):
    python profiles/batch_sweep.py [f32|f16] [size]
Prints one line per batch: N, ms per forward (median of 5 groups of forwards), images/s, and which 3x3 launches walk."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import celebrity_image_denoiser_amd as cid
from celebrity_image_denoiser_amd import synth

dtype = sys.argv[1] if len(sys.argv) > 1 else "f32"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
m = cid.load(synth.make_state_dict("default"), device="cuda:0", strict=True)
m.compute_dtype = dtype
base, _, _ = synth.make_batch(16, S, S, first_index=0)
xb = torch.from_numpy(base).to("cuda:0")
print(f"# dtype {dtype}, {S}x{S}x3, default weights")
for n in (1, 2, 4, 8, 16, 24, 32, 48, 64, 96, 128, 192, 256, 384, 512, 768, 1024):
    x = xb.repeat((n + 15) // 16, 1, 1, 1)[:n].contiguous()
    reps = max(3, min(200, int(4096 / n)))
    for _ in range(3):
        m(x)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(reps):
            m(x)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / reps)
    t = sorted(ts)[2]
    print(f"N={n:5d}  {t * 1e3:9.4f} ms  {n / t:10.1f} images/s", flush=True)
    del x
    m._ws = None
    torch.cuda.empty_cache()
