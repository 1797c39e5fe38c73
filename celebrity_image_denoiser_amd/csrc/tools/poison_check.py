"""Ad-hoc: does any kernel read arena memory that no kernel of the same forward wrote?  Run a forward (arena = reference values), poison
the arena with NaN, run again, and report per arena buffer (dataflow order) how many elements differ from the reference."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../../..")
import celebrity_image_denoiser_amd as cid
from celebrity_image_denoiser_amd import synth
N, H, W = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (192, 128, 128)))
algo = sys.argv[4] if len(sys.argv) > 4 else "winograd42"
dtype = sys.argv[5] if len(sys.argv) > 5 else "f32"
m = cid.load(synth.make_state_dict("hot"), device="cuda:0", strict=True)
m.conv_algo = algo
m.compute_dtype = dtype
x = (torch.rand((N, 3, H, W), device="cuda:0") * 2 - 1).contiguous()
y_ref = m(x).clone(); torch.cuda.synchronize()
ref = m._ws.clone()
H1, W1 = H // 2, W // 2; H2, W2 = H1 // 2, W1 // 2; Hu2, Wu2 = 2 * H2, 2 * W2; Hu1, Wu1 = 2 * Hu2, 2 * Wu2
s0, s1, s2, su2, su1 = N * H * W, N * H1 * W1, N * H2 * W2, N * Hu2 * Wu2, N * Hu1 * Wu1
names = ["t0", "cat1", "p1", "t1", "cat2", "p2", "t2", "bt", "t3", "d2", "t4"]
sizes = [s0 * 64, su1 * 128, s1 * 64, s1 * 128, su2 * 256, s2 * 128, s2 * 256, s2 * 256, su2 * 128, su2 * 128, su1 * 64]
off, o = [], 0
for sz in sizes: off.append(o); o = (o + sz + 63) // 64 * 64
wsf = m._ws.view(torch.float32)
wsf.fill_(float("nan")); torch.cuda.synchronize()
from celebrity_image_denoiser_amd import _lib
assert _lib.lib().cid_debug_poison_lds(torch.cuda.current_stream().cuda_stream) == 0
y = m(x); torch.cuda.synchronize()
reff = ref.view(torch.float32)
print(f"{algo} {dtype} N={N} {H}x{W}: output equal to reference: {torch.equal(y, y_ref)}; NaNs in output: {int(torch.isnan(y).sum())}")
order = ["t0", "cat1", "p1", "t1", "cat2", "p2", "t2", "bt", "t3", "d2", "t4"]
for nm, of, sz in (zip(names, off, sizes) if dtype == 'f32' else []):
    a, b = wsf[of:of + sz], reff[of:of + sz]
    nan = int(torch.isnan(a).sum()); diff = int(((a != b) & ~torch.isnan(a)).sum())
    print(f"  {nm:5s} elements {sz:12d}  still NaN (never written) {nan:10d}  written but different {diff:10d}")
# where do the t0 differences sit?
a, b = wsf[off[0]:off[0] + sizes[0]].view(N, H, W, 64), reff[off[0]:off[0] + sizes[0]].view(N, H, W, 64)
d = (a != b).nonzero() if dtype == 'f32' else torch.zeros(0)
if d.numel():
    print("t0 differing elements:", d.shape[0], " first few (n,y,x,c):", d[:8].tolist())
    print("  distinct x:", sorted(set(d[:, 2].tolist()))[:40], " distinct y%8:", sorted(set((d[:, 1] % 8).tolist())), " distinct c:", len(set(d[:, 3].tolist())))
    print("  values now/ref:", a[tuple(d[0])].item(), b[tuple(d[0])].item())
