"""Ad-hoc: stray-write check.  The arena, the input and the output sit inside larger buffers filled with a sentinel; after
forwards of several shapes/algorithms/dtypes the guard zones on both sides must be untouched."""
import sys, os, ctypes, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../../..")
import celebrity_image_denoiser_amd as cid
from celebrity_image_denoiser_amd import synth, _lib
G = 1 << 20   # guard bytes on each side
m = cid.load(synth.make_state_dict("hot"), device="cuda:0", strict=True)
bad = 0
for algo in ("winograd42", "winograd64", "direct", "split16"):
    for dtype in (("f32",) if algo == "split16" else ("f32", "f16")):
        m.conv_algo, m.compute_dtype = algo, dtype
        for (n, h, w) in ((5, 128, 128), (3, 40, 72), (2, 21, 30), (1, 4, 4), (2, 130, 31), (600, 128, 128)):
            need = ctypes.c_size_t(); _lib.check(m._cid, _lib.lib().cid_workspace_bytes(n, h, w, ctypes.byref(need)))
            big = torch.full((need.value + 2 * G,), 0xA5, dtype=torch.uint8, device="cuda:0")
            m._ws = big[G:G + need.value]
            ho, wo = 4 * (h // 4), 4 * (w // 4)
            xbig = torch.full((n * 3 * h * w + 2 * G // 4,), 7.0, device="cuda:0")
            x = xbig[G // 4:G // 4 + n * 3 * h * w].view(n, 3, h, w); x.uniform_(-1, 1)
            ybig = torch.full((n * 3 * ho * wo + 2 * G // 4,), 9.0, device="cuda:0")
            y = ybig[G // 4:G // 4 + n * 3 * ho * wo].view(n, 3, ho, wo)
            m(x, out=y)
            u = (torch.rand((n, h, w, 3), device="cuda:0") * 255).to(torch.uint8)
            y8big = torch.full((n * ho * wo * 3 + 2 * G,), 0x5A, dtype=torch.uint8, device="cuda:0")
            y8 = y8big[G:G + n * ho * wo * 3].view(n, ho, wo, 3)
            m.forward_u8(u, out=y8)
            torch.cuda.synchronize()
            ok = bool((big[:G] == 0xA5).all() and (big[G + need.value:] == 0xA5).all() and (xbig[:G // 4] == 7.0).all() and (xbig[G // 4 + n * 3 * h * w:] == 7.0).all()
                      and (ybig[:G // 4] == 9.0).all() and (ybig[G // 4 + n * 3 * ho * wo:] == 9.0).all() and (y8big[:G] == 0x5A).all() and (y8big[G + n * ho * wo * 3:] == 0x5A).all())
            if not ok:
                bad += 1; print("GUARD VIOLATION", algo, dtype, (n, h, w))
            m._ws = None
print("guard violations:", bad)
sys.exit(1 if bad else 0)
