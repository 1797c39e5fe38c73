"""Ad-hoc: a batch far beyond the benchmark's (index arithmetic past 2^31 elements per buffer), checked against small-batch runs."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../../..")
import celebrity_image_denoiser_amd as cid
from celebrity_image_denoiser_amd import synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
m = cid.load(synth.make_state_dict("hot"), device="cuda:0", strict=True)
base, _, _ = synth.make_batch(16, 128, 128, first_index=7000)
x = torch.from_numpy(base).to("cuda:0").repeat((N + 15) // 16, 1, 1, 1)[:N].contiguous()
y = m(x); torch.cuda.synchronize()
ref = m(x[:16].contiguous()); torch.cuda.synchronize()
bad = 0
for k in range(0, N, 16):
    n = min(16, N - k)
    if not torch.equal(y[k:k + n], ref[:n]): bad += 1
print(f"N={N}: arena {m._ws.numel()/2**30:.1f} GiB, t0 elements {N*128*128*64:.3e}, mismatching groups: {bad}")
assert bad == 0
for dt in ("f16",):
    m.compute_dtype = dt
    y = m(x); ref = m(x[:16].contiguous()); torch.cuda.synchronize()
    bad = sum(0 if torch.equal(y[k:k + min(16, N - k)], ref[:min(16, N - k)]) else 1 for k in range(0, N, 16))
    print(f"{dt}: mismatching groups: {bad}"); assert bad == 0
u8 = torch.from_numpy(synth.make_batch(16, 128, 128, first_index=7000)[2]).to("cuda:0").repeat((N + 15) // 16, 1, 1, 1)[:N].contiguous()
m.compute_dtype = "f32"
y8 = m.forward_u8(u8); r8 = m.forward_u8(u8[:16].contiguous()); torch.cuda.synchronize()
bad = sum(0 if torch.equal(y8[k:k + min(16, N - k)], r8[:min(16, N - k)]) else 1 for k in range(0, N, 16))
print(f"u8: mismatching groups: {bad}"); assert bad == 0
print("ok")
