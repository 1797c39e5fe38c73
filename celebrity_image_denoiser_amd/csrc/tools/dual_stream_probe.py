"""Ad-hoc: does running two half-batches on two HIP streams (kernels of different layers overlapping on the chip) beat one
full batch on one stream?  Two module instances (own arenas), same weights."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../../..")
import celebrity_image_denoiser_amd as cid
from celebrity_image_denoiser_amd import synth
sd = synth.make_state_dict("default")
dev = torch.device("cuda:0")
x = torch.from_numpy(synth.make_batch(256, 128, 128)[0]).to(dev)
def bench(fn, steps=40, warm=8):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e3
m = cid.load(sd, device=dev, strict=True)
for rnd in range(2):
    t1 = bench(lambda: m(x))
    for parts in (2, 4):
        ms = [cid.load(sd, device=dev, strict=True) for _ in range(parts)]
        ss = [torch.cuda.Stream(dev) for _ in range(parts)]
        xs = [c.contiguous() for c in x.chunk(parts)]
        def run():
            for mm, s, xx in zip(ms, ss, xs):
                with torch.cuda.stream(s):
                    mm(xx)
        tp = bench(run)
        print(f"round {rnd}: one stream {t1:.3f} ms ({256/t1*1e3:.0f} img/s)   {parts} streams x {256//parts} images {tp:.3f} ms ({256/tp*1e3:.0f} img/s)  ratio {t1/tp:.4f}")
        del ms
