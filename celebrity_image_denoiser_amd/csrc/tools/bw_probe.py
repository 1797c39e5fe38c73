"""HBM ceilings seen by simple streaming kernels on this box (context for the head/tail roofline fractions)."""
import torch, time
d = "cuda:0"
n = 256 * 128 * 128 * 64          # floats: the head's output (1.07 GB)
a = torch.empty(n, device=d); b = torch.empty(n, device=d)
def t(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
ms = t(lambda: a.fill_(1.0)); print(f"fill   1.07 GB write        {ms:.4f} ms  {n*4/ms/1e6:.0f} GB/s")
ms = t(lambda: b.copy_(a)); print(f"copy   1.07 GB read + write {ms:.4f} ms  {2*n*4/ms/1e6:.0f} GB/s")
ms = t(lambda: a.sum()); print(f"sum    1.07 GB read         {ms:.4f} ms  {n*4/ms/1e6:.0f} GB/s")
ms = t(lambda: torch.relu_(a)); print(f"relu_  1.07 GB read + write {ms:.4f} ms  {2*n*4/ms/1e6:.0f} GB/s")
