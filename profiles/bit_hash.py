#!/usr/bin/env python3
"""Print SHA-256 digests of the forward's outputs for a fixed set of inputs, repeated several times (run-to-run stability), with the
library named by CID_LIB_PATH.  Two builds that execute the same arithmetic must print the same digests:
    CID_LIB_PATH=build_ab/libcid_x.so python profiles/bit_hash.py [repeats]"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import celebrity_image_denoiser_amd as cid
from celebrity_image_denoiser_amd import synth
from oracle import torch_oracle

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for wset in ("default", "hot"):
    sd = synth.make_state_dict(wset)
    m = cid.load(sd, device="cuda:0", strict=True)
    for (n, s, first) in ((256, 128, 0), (256, 128, 4000), (24, 256, 300), (7, 52, 77)):
        x_host, _, _ = synth.make_batch(n, s, s, first_index=first)
        x = torch.from_numpy(x_host).to("cuda:0")
        digs = set()
        for r in range(reps):
            y = m(x)
            torch.cuda.synchronize()
            digs.add(hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest()[:16])
        err = float(np.abs(y[:2].cpu().numpy() - torch_oracle.forward(sd, x_host[:2]).numpy()).max())
        print(wset, n, s, first, "digests:", sorted(digs), "stable" if len(digs) == 1 else "UNSTABLE", "err_vs_oracle %.3e" % err, flush=True)
