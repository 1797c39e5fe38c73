"""Ad-hoc: many large random batches through both Winograd paths (fused and unfused last layer) and the direct path; a
synchronisation race (LDS-DMA waits, barriers, LDS reuse in the epilogues) would show up as a bit difference between two runs of
one configuration, or as a difference beyond the fp32 tolerance between configurations."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/../../..")
import celebrity_image_denoiser_amd as cid
from celebrity_image_denoiser_amd import synth
m = cid.load(synth.make_state_dict("hot"), device="cuda:0", strict=True)
g = torch.Generator(device="cuda:0"); g.manual_seed(1234)
bad = 0
worst_h16 = worst_t16 = 0.0
for it in range(40):
    n, hw = (192, 128) if it % 2 == 0 else (96, int(torch.randint(33, 200, (1,)).item()))
    x = (torch.rand((n, 3, hw, hw + (it % 5) * 3), device="cuda:0", generator=g) * 2 - 1).contiguous()
    m.conv_algo, m.tail_algo = "winograd42", "fused"; a = m(x).clone(); b = m(x).clone()
    m.tail_algo = "bands"; c = m(x).clone(); c2 = m(x).clone()
    m.conv_algo, m.tail_algo = "winograd64", "fused"; e = m(x).clone(); e2 = m(x).clone()
    m.conv_algo, m.tail_algo = "direct", "tiles"; d = m(x).clone()
    # the opt-in split-operand algorithm (fused last layer, two column blocks per workgroup, transposed convolutions included): run to run bit-equal, within 2e-5 of the default
    m.conv_algo, m.tail_algo = "split16", "fused"; s1 = m(x).clone(); s2 = m(x).clone()
    torch.cuda.synchronize()
    worst_s16 = max(globals().get("worst_s16", 0.0), float((s1 - a).abs().max()))
    if not (torch.equal(s1, s2) and float((s1 - a).abs().max()) <= 2e-5):
        bad += 1; print("split16 MISMATCH at iteration", it, tuple(x.shape), float((s1 - s2).abs().max()), float((s1 - a).abs().max()))
    # two runs of one configuration: bit-equal.  Two ALGORITHMS: each is within the 1e-5 contract of the exact result, so within 2e-5 of each
    # other (white-noise inputs on the He-gain weights are the worst case: F(4x2) against the direct kernel has reached 1.03e-5 here)
    if not (torch.equal(a, b) and torch.equal(c, c2) and torch.equal(e, e2) and float((a - c).abs().max()) <= 1e-5
            and float((a - d).abs().max()) <= 2e-5 and float((a - e).abs().max()) <= 2e-5):
        bad += 1; print("MISMATCH at iteration", it, tuple(x.shape), float((a - b).abs().max()), float((a - c).abs().max()), float((a - d).abs().max()), float((a - e).abs().max()))
    # the fp16-storage path (round 4: results stored straight from the accumulators; last layer fused through a wave-private, XOR-swizzled LDS
    # staging area, or as its own tiled kernel): two runs of one form bit-equal; the two forms within what z's rounding to half costs; both
    # within the path's stated 5e-3 of the fp32 result (white noise on He-gain weights is its worst case too)
    m.conv_algo, m.tail_algo, m.compute_dtype = "winograd42", "fused", "f16"
    h1 = m(x).clone(); h2 = m(x).clone()
    m.tail_algo = "tiles"; t1 = m(x).clone(); t2 = m(x).clone()
    m.tail_algo, m.compute_dtype = "fused", "f32"
    torch.cuda.synchronize()
    # (white noise on He-gain weights lies outside the input distribution the 5e-3 contract is stated on — SURVEY 8d: smooth images + sigma = 25 noise —
    # and exceeds it in EITHER form over ~10^7 outputs: the bound checked here is 8e-3, and both distances are printed)
    worst_h16 = max(worst_h16, float((h1 - a).abs().max())); worst_t16 = max(worst_t16, float((t1 - a).abs().max()))
    if not (torch.equal(h1, h2) and torch.equal(t1, t2) and float((h1 - t1).abs().max()) <= 2e-3 and float((h1 - a).abs().max()) <= 8e-3):
        bad += 1; print("fp16 MISMATCH at iteration", it, tuple(x.shape), float((h1 - h2).abs().max()), float((t1 - t2).abs().max()), float((h1 - t1).abs().max()), float((h1 - a).abs().max()))
print("fp16 storage vs fp32 on white noise, worst over all iterations: fused last layer %.3e, separate last layer %.3e" % (worst_h16, worst_t16))
print("split16 vs winograd42 on white noise, worst over all iterations: %.3e" % globals().get("worst_s16", 0.0))
print("iterations with a mismatch:", bad)
sys.exit(1 if bad else 0)
