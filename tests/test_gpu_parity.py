"""GPU parity: the HIP forward, called through the C ABI (via the Python host class), against
  (1) the golden fixtures recorded from the reference class itself, and
  (2) the CPU oracles (ATen restatement = the reference's arithmetic provider; C restatement).

Stated fp32 tolerances (BASELINE.md section 4 / SURVEY 8d):
    per-pixel  max|y_hip - y_ref| <= 1e-5      on the tanh output
    per-stage  max|delta|         <= 1e-5 * max(1, max|stage|)
    PSNR_delta = |PSNR(y_hip, clean) - PSNR(y_ref, clean)| <= 0.01 dB
"""
import glob
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from celebrity_image_denoiser_amd import synth  # noqa: E402

TOL = 1e-5


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a real MI355X (torch.cuda.is_available() is False)")


def _default_tail(algo):
    """The last-layer form each 3x3 algorithm is tested with by default: the fused contraction needs a Winograd kernel in front of it."""
    return {"direct": "tiles"}.get(algo, "fused")


@pytest.fixture(scope="module", params=["winograd42", "winograd64", "direct", "split16"])
def models(request, weight_sets):
    """All four algorithms of the 3x3 GEMM layers go through every parity test: Winograd F(4x2,3x3) (the default),
    Winograd F(2x2,3x3), the 9-tap implicit GEMM and (round 4, opt-in) the split-operand form on the fp16 MFMA — same tensors,
    same 1e-5 tolerance; and with them the decompositions of the last layer: fused into upconv1[0]'s epilogue (default),
    the row-band kernel and the tiled kernel."""
    _need_gpu()
    import celebrity_image_denoiser_amd as cid

    out = {}
    for k, v in weight_sets.items():
        m = cid.load(v, device="cuda:0", strict=True)
        m.conv_algo = request.param
        m.tail_algo = _default_tail(request.param)
        assert m.conv_algo == request.param and m.tail_algo == _default_tail(request.param)
        out[k] = m
    return out


def _run(model, x):
    y = model(torch.from_numpy(np.ascontiguousarray(x)).to("cuda:0"))
    torch.cuda.synchronize()
    return y.cpu().numpy()


@pytest.mark.parametrize("name", sorted(os.path.basename(p)[:-4] for p in glob.glob(
    os.path.join(os.path.dirname(__file__), "golden", "tiny_*.npz"))))
def test_golden_tiny(models, golden_dir, name):
    """16x16, 20x24 (ragged tiles), 13x18 / 7x9 (crop path: output 12x16 / 4x8), 4x4 (minimum)."""
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    y = _run(models[name.split("_")[1]], g["x"])
    assert y.shape == g["out"].shape
    assert np.abs(y - g["out"]).max() <= TOL


@pytest.mark.parametrize("wset", ["default", "hot"])
def test_golden_128(models, golden_dir, weight_sets, wset):
    from celebrity_image_denoiser_amd import psnr

    g = np.load(os.path.join(golden_dir, f"full_{wset}_128.npz"))
    x, clean, _ = synth.make_batch(2, 128, 128, 100)
    y = _run(models[wset], x)
    assert np.abs(y - g["out"]).max() <= TOL
    assert abs(psnr(y, clean) - psnr(g["out"], clean)) <= 0.01


@pytest.mark.parametrize("wset", ["default", "hot"])
@pytest.mark.parametrize("case", ["n4_128", "n1_256"])
def test_golden_stats(models, golden_dir, wset, case):
    """Output moments / sampled pixels / PSNR of the reference at N=4 128x128 and N=1 256x256."""
    from celebrity_image_denoiser_amd import psnr

    st = json.load(open(os.path.join(golden_dir, "stats.json")))[f"{wset}_{case}"]
    x, clean, _ = synth.make_batch(st["n"], st["h"], st["w"], st["first_index"])
    y = _run(models[wset], x)
    g = st["stages"]["out"]
    assert list(y.shape) == g["shape"]
    assert np.abs(y.reshape(-1)[g["idx"]] - np.array(g["samples"])).max() <= TOL
    assert abs(float(y.astype(np.float64).sum()) - g["sum"]) <= 1e-5 * y.size
    assert abs(psnr(y, clean) - st["psnr_out_vs_clean"]) <= 0.01


@pytest.mark.parametrize("wset", ["default", "hot"])
@pytest.mark.parametrize("shape", [(3, 40, 72), (1, 64, 64), (5, 36, 100), (2, 9, 4)])
def test_against_cpu_oracles(models, weight_sets, wset, shape):
    """Seeded inputs at sizes the oracles finish in seconds, incl. multi-tile and ragged shapes."""
    from oracle import c_oracle, torch_oracle

    n, h, w = shape
    x, _, _ = synth.make_batch(n, h, w, first_index=500)
    y = _run(models[wset], x)
    ref_t = torch_oracle.forward(weight_sets[wset], x).numpy()
    ref_c = c_oracle.forward(weight_sets[wset], x)
    assert y.shape == ref_t.shape
    assert np.abs(y - ref_t).max() <= TOL
    assert np.abs(y - ref_c).max() <= TOL


def test_closer_to_exact_than_tolerance(models, weight_sets):
    """Against the all-float64 forward the HIP result must be as accurate as the fp32 reference is
    (both are fp32 evaluations of the same real-valued function)."""
    from oracle import torch_oracle

    x, _, _ = synth.make_batch(2, 64, 64, first_index=700)
    exact = torch_oracle.forward(weight_sets["hot"], x, dtype=torch.float64).numpy()
    ref32 = torch_oracle.forward(weight_sets["hot"], x).numpy()
    y = _run(models["hot"], x)
    e_hip, e_ref = np.abs(y - exact).max(), np.abs(ref32 - exact).max()
    assert e_hip <= max(4 * e_ref, 2e-6)


def test_batch_independence_full_size(models):
    """Size-independent property at BASELINE config-2 image size: every image of a batch of 24
    equals the same image run alone, bit for bit (the kernels have no cross-sample term)."""
    x, _, _ = synth.make_batch(24, 128, 128, first_index=1000)
    yb = _run(models["hot"], x)
    for i in (0, 7, 23):
        assert np.array_equal(yb[i:i + 1], _run(models["hot"], x[i:i + 1]))


def test_deterministic(models):
    x, _, _ = synth.make_batch(8, 128, 128, first_index=1100)
    assert np.array_equal(_run(models["default"], x), _run(models["default"], x))


def test_zero_weights_give_tanh_bias(models):
    """Algebraic property: with all conv weights zero the output is tanh(bias of the last conv)."""
    import celebrity_image_denoiser_amd as cid

    sd = {k: np.zeros_like(v) for k, v in synth.make_state_dict("default").items()}
    sd["upconv1.2.bias"] = np.array([0.25, -0.5, 1.5], np.float32)
    m = cid.load(sd, device="cuda:0", strict=True)
    y = _run(m, synth.make_batch(1, 16, 24)[0])
    for c in range(3):
        assert np.allclose(y[0, c], np.tanh(sd["upconv1.2.bias"][c]), atol=1e-6)


def test_errors_are_loud(models):
    m = models["default"]
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 8, 8))                      # CPU tensor: no fallback
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 3, 3, device="cuda:0"))     # too small, like the reference
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 8, 8, device="cuda:0", dtype=torch.float16))
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 4, 8, 8, device="cuda:0"))


def test_state_dict_reload_repacks(models, weight_sets):
    """load_state_dict on a live module takes effect on the next forward (weights repacked)."""
    import celebrity_image_denoiser_amd as cid

    x, _, _ = synth.make_batch(1, 16, 16)
    m = cid.load(weight_sets["default"], device="cuda:0")
    m.conv_algo, m.tail_algo = models["hot"].conv_algo, models["hot"].tail_algo
    y0 = _run(m, x)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in weight_sets["hot"].items()}, strict=True)
    y1 = _run(m, x)
    assert np.array_equal(y1, _run(models["hot"], x)) and not np.array_equal(y0, y1)


STAGES = ("down1", "pool1", "down2", "pool2", "bottleneck", "up2", "upconv2", "up1")


@pytest.mark.parametrize("name", sorted(os.path.basename(p)[:-4] for p in glob.glob(
    os.path.join(os.path.dirname(__file__), "golden", "tiny_*.npz"))))
def test_every_stage_against_the_reference_hooks(models, golden_dir, name):
    """Per-stage parity (only `out` was compared, so a regression could not be localised and compensating
    errors in intermediates were invisible).  The tiny fixtures hold what forward hooks on the reference module's
    submodules recorded; the same tensors are read back from the activation arena (cid_stage_view) after the forward.
    Skip tensors are stored only over the top-left region the concat keeps (13x18 -> 12x16, 7x9 -> 4x8).
    Stated tolerance: max|delta| <= 1e-5 * max(1, max|stage|)."""
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    m = models[name.split("_")[1]]
    n, _, h, w = g["x"].shape
    y = _run(m, g["x"])
    assert np.abs(y - g["out"]).max() <= TOL
    # `upconv1` (pre-tanh) is fused into the last kernel and never stored: out = tanh(upconv1) ties it down
    assert np.abs(y - np.tanh(g["upconv1"])).max() <= TOL
    for st in STAGES:
        got = m.stage_output(st, n, h, w).cpu().numpy()
        ref = g[st][:, :, :got.shape[2], :got.shape[3]]
        assert got.shape == ref.shape and got.shape[1] == g[st].shape[1], (st, got.shape, g[st].shape)
        if st not in ("down1", "down2"):
            assert got.shape == g[st].shape, (st, got.shape, g[st].shape)    # only skip tensors are cropped
        tol = TOL * max(1.0, float(np.abs(ref).max()))
        assert np.abs(got - ref).max() <= tol, (st, float(np.abs(got - ref).max()), tol)
    with pytest.raises(KeyError):
        m.stage_output("upconv1", n, h, w)


@pytest.mark.parametrize("wset", ["default", "hot"])
def test_iterated_denoise_golden(models, golden_dir, wset):
    """the iterated caller feeds the output back three times (denoise_eavl_iter.py:93-96).  Fixture:
    the class of THAT file (lifted like app.py's) iterated on the CPU; it also records how far the reference's own fp32
    run drifts from the float64 iteration (hot weights: 1.0e-6, 1.8e-6, 2.4e-6), so 1e-5 holds at every iteration."""
    import celebrity_image_denoiser_amd as cid
    from celebrity_image_denoiser_amd import HostPipeline
    from celebrity_image_denoiser_amd.api import to_unit_range

    g = np.load(os.path.join(golden_dir, f"iter3_{wset}_32x32.npz"))
    assert float(g["fp32_vs_fp64_maxabs"].max()) < TOL / 3
    m = models[wset]
    x = torch.from_numpy(g["x"])
    for k in (1, 2, 3):
        y = cid.denoise(m, x, iterations=k)
        assert y.device.type == "cpu" and y.shape == x.shape
        assert np.abs(y.numpy() - g[f"iter{k}"]).max() <= TOL, k
    y3 = cid.denoise(m, x, iterations=3)
    yp = HostPipeline(m)([g["x"]], iterations=3)[0]                  # upload / 3 forwards on the device / download
    assert torch.equal(yp, y3)
    assert torch.equal(cid.denoise(m, x, iterations=3, max_batch=2), y3)
    # the saved view (:108-109): current*0.5+0.5 through ToPILImage (truncating)
    u8 = to_unit_range(y3).mul(255).byte().permute(0, 2, 3, 1).numpy()
    d = np.abs(u8.astype(np.int16) - g["final_u8"].astype(np.int16))
    assert d.max() <= 1 and (d != 0).mean() <= 1e-3
    # ... which the product takes with one HIP kernel (cid_view_u8, harness.enhance_images): the same bytes as the expression above,
    # and as the uint8 output format of the forward that produced y3 (cid_forward_ex) would have held
    y3d = y3.to("cuda:0")
    assert np.array_equal(m.view_u8(y3d).cpu().numpy(), u8)
    y2d = cid.denoise(m, x, iterations=2).to("cuda:0")
    assert torch.equal(m.view_u8(m(y2d)), m.forward_fmt(y2d, out_u8=True))
    edge = torch.tensor([-1.5, -1.0, -0.999, -0.5, 0.0, 0.25, 0.999, 1.0, 1.5, float("nan")] * 12, device="cuda:0").view(1, 3, 5, 8)   # clamp ends, NaN -> 0 like fmaxf/fminf
    got = m.view_u8(edge).cpu().numpy()
    ok = ~torch.isnan(edge).cpu().numpy().transpose(0, 2, 3, 1)
    assert np.array_equal(got[ok], to_unit_range(edge).mul(255).byte().permute(0, 2, 3, 1).cpu().numpy()[ok])
    with pytest.raises(RuntimeError):
        m.view_u8(y3)                                               # CPU tensor: no fallback


def test_load_checkpoint_by_path_and_run(tmp_path, golden_dir, weight_sets):
    """SURVEY 8f row f3 on the GPU: cid.load(path) = torch-free reader -> load_state_dict -> pack -> forward, against the
    golden output, for the trainer's checkpoint layout (training.py:359-376), the DataParallel 'module.'-prefixed keys the
    reference loader strips (app.py:269-271), a bare state_dict, and the legacy (pre-zip) format."""
    _need_gpu()
    import celebrity_image_denoiser_amd as cid

    g = np.load(os.path.join(golden_dir, "tiny_hot_20x24.npz"))
    sd = {k: torch.from_numpy(v) for k, v in weight_sets["hot"].items()}
    extra = {"discriminator": {"w": torch.ones(3, 3)}, "g_optimizer": {"state": {}, "param_groups": [{"lr": 1e-4, "betas": (0.5, 0.999)}]},
             "epoch": 499, "best_psnr": 30.5, "metric_history": {"psnr": [1.0]}}
    files = {
        "trainer.pth": dict(generator=sd, **extra),
        "dataparallel.pth": dict(generator={"module." + k: v for k, v in sd.items()}, **extra),
        "state_dict_key.pth": {"state_dict": sd},
        "bare.pth": sd,
    }
    for fname, obj in files.items():
        path = os.path.join(tmp_path, fname)
        torch.save(obj, path)
        m = cid.load(path, device="cuda:0", strict=True)
        assert np.abs(_run(m, g["x"]) - g["out"]).max() <= TOL, fname
        assert all(torch.equal(m.state_dict()[k].cpu(), v) for k, v in sd.items()), fname
    legacy = os.path.join(tmp_path, "legacy.pth")
    torch.save(files["dataparallel.pth"], legacy, _use_new_zipfile_serialization=False)
    assert np.abs(_run(cid.load(legacy, device="cuda:0", strict=True), g["x"]) - g["out"]).max() <= TOL
    # the reference's own loader entry point (app.py:257-274) on the build's module
    m = cid.DenoiseGenerator().to("cuda:0")
    cid.load_state_safely(m, os.path.join(tmp_path, "dataparallel.pth"))
    assert not m.training and np.abs(_run(m, g["x"]) - g["out"]).max() <= TOL
    with pytest.raises(RuntimeError):                                   # strict: a missing tensor is an error
        torch.save({"generator": {k: v for k, v in sd.items() if k != "up1.bias"}}, os.path.join(tmp_path, "short.pth"))
        cid.load(os.path.join(tmp_path, "short.pth"), device="cuda:0", strict=True)


def test_directory_harness_like_the_reference_eval_scripts(tmp_path, weight_sets):
    """`enhance_images` = the reference's denoisegan_eval.py:62-103 / denoise_eavl_iter.py:62-114 loop on the GPU pipeline:
    PNG/JPEG files in, bicubic resize (PIL, as in the reference), network, `*0.5+0.5`, ToPILImage truncation, files out
    under the reference's names.  Checked against the CPU oracle fed with the same PIL-resized images."""
    _need_gpu()
    from PIL import Image

    import celebrity_image_denoiser_amd as cid
    from oracle import torch_oracle

    src, dst, dst3 = tmp_path / "testNoise", tmp_path / "testOp", tmp_path / "testOp3"
    src.mkdir()
    _, _, noisy = synth.make_batch(5, 40, 52, first_index=8000)
    for k in range(5):
        Image.fromarray(noisy[k]).save(src / f"img{k}.{'png' if k % 2 == 0 else 'jpg'}")
    (src / "notes.txt").write_text("not an image")
    (src / "broken.png").write_bytes(b"not a png")
    ckpt = tmp_path / "denoise_epoch_499.pth"
    torch.save({"generator": {k: torch.from_numpy(v) for k, v in weight_sets["hot"].items()}, "epoch": 499}, ckpt)

    written = cid.enhance_images(str(ckpt), str(src), str(dst), image_size=(32, 48), batch_size=3)
    assert sorted(os.path.basename(p) for p in written) == [f"img{k}.{'png' if k % 2 == 0 else 'jpg'}" for k in range(5)]
    for k in (0, 2, 4):   # PNG outputs are lossless: compare with the oracle on the same resized input
        with Image.open(src / f"img{k}.png") as im:
            x_u8 = np.asarray(im.convert("RGB").resize((32, 48), resample=Image.Resampling.BICUBIC))
        ref = torch_oracle.forward(weight_sets["hot"], synth.normalize_u8(x_u8[None]))
        ref_u8 = (ref * 0.5 + 0.5).clamp(0, 1).mul(255).byte().permute(0, 2, 3, 1).numpy()[0]
        got = np.asarray(Image.open(dst / f"img{k}.png"))
        d = np.abs(got.astype(np.int16) - ref_u8.astype(np.int16))
        assert got.shape == (48, 32, 3) and d.max() <= 1 and (d != 0).mean() <= 2e-3
    # iterated variant: <base>_iter<i><ext> and <base>_final<ext>, three passes on the device
    m = cid.load(str(ckpt), device="cuda:0", strict=True)
    written3 = cid.enhance_images(None, str(src), str(dst3), image_size=(32, 48), num_iterations=3, model=m)
    assert len(written3) == 5 * 4 and os.path.exists(dst3 / "img0_iter1.png") and os.path.exists(dst3 / "img0_final.png")
    assert np.array_equal(np.asarray(Image.open(dst3 / "img0_iter3.png")), np.asarray(Image.open(dst3 / "img0_final.png")))
    assert np.array_equal(np.asarray(Image.open(dst3 / "img0_iter1.png")), np.asarray(Image.open(dst / "img0.png")))
    with Image.open(src / "img0.png") as im:
        x_u8 = np.asarray(im.convert("RGB").resize((32, 48), resample=Image.Resampling.BICUBIC))
    z = torch.from_numpy(synth.normalize_u8(x_u8[None]))
    for _ in range(3):
        z = torch_oracle.forward(weight_sets["hot"], z)
    ref_u8 = (z * 0.5 + 0.5).clamp(0, 1).mul(255).byte().permute(0, 2, 3, 1).numpy()[0]
    d = np.abs(np.asarray(Image.open(dst3 / "img0_final.png")).astype(np.int16) - ref_u8.astype(np.int16))
    assert d.max() <= 1 and (d != 0).mean() <= 5e-3


def test_host_roundtrip_and_batch_split(models):
    import celebrity_image_denoiser_amd as cid

    x = torch.from_numpy(synth.make_batch(3, 32, 32)[0])
    y_split = cid.denoise(models["hot"], x, max_batch=2)
    assert y_split.device.type == "cpu" and torch.equal(y_split, cid.denoise(models["hot"], x))


@pytest.mark.parametrize("wset", ["default", "hot"])
def test_row_band_tail_kernel(weight_sets, golden_dir, wset):
    """The separate row-band kernel for the last layer (tail_algo="bands": what the fused default falls back to when the
    3x3 layers run as direct GEMMs): golden fixtures incl. the crop/ragged ones, a multi-band batch, uint8 output."""
    _need_gpu()
    import celebrity_image_denoiser_amd as cid

    m = cid.load(weight_sets[wset], device="cuda:0", strict=True)
    m.tail_algo = "bands"
    assert m.tail_algo == "bands"
    for name in ("16x16", "20x24", "13x18", "7x9", "4x4"):
        g = np.load(os.path.join(golden_dir, f"tiny_{wset}_{name}.npz"))
        assert np.abs(_run(m, g["x"]) - g["out"]).max() <= TOL, name
    g = np.load(os.path.join(golden_dir, f"full_{wset}_128.npz"))
    x, _, noisy = synth.make_batch(2, 128, 128, 100)
    assert np.abs(_run(m, x) - g["out"]).max() <= TOL
    gu = np.load(os.path.join(golden_dir, f"u8_{wset}_32x40.npz"))
    d = np.abs(m.forward_u8(torch.from_numpy(gu["noisy_u8"]).to("cuda:0")).cpu().numpy().astype(np.int16) - gu["out_u8"].astype(np.int16))
    assert d.max() <= 1 and (d != 0).mean() <= 1e-3
    xb, _, _ = synth.make_batch(9, 100, 72, first_index=3300)          # several bands per image, ragged width
    ref = cid.load(weight_sets[wset], device="cuda:0", strict=True)
    ref.tail_algo = "tiles"
    assert np.abs(_run(m, xb) - _run(ref, xb)).max() <= TOL


def test_winograd_and_direct_agree(weight_sets):
    """The two algorithms are independent fp32 evaluations of the same convolutions."""
    _need_gpu()
    import celebrity_image_denoiser_amd as cid

    x, _, _ = synth.make_batch(4, 128, 128, first_index=1300)
    m = cid.load(weight_sets["hot"], device="cuda:0", strict=True)
    assert m.conv_algo == "winograd42"   # the default
    yw = _run(m, x)
    m.conv_algo = "direct"
    yd = _run(m, x)
    assert np.abs(yw - yd).max() <= TOL and not np.array_equal(yw, yd)


def test_adopt_device_blob_like_a_broadcast_receiver(weight_sets):
    """What a non-source rank does after the RCCL broadcast: attach a packed device blob produced by another
    module, refresh the nn.Parameters from it, compute the same outputs."""
    _need_gpu()
    import celebrity_image_denoiser_amd as cid

    src = cid.load(weight_sets["hot"], device="cuda:0", strict=True)
    dst = cid.load(None, device="cuda:0")                      # random init, like a rank that read no checkpoint
    blob = src.pack_weights().clone()
    dst.adopt_packed_weights(blob, update_parameters=True)
    assert all(torch.equal(dst.state_dict()[k].cpu(), torch.from_numpy(v)) for k, v in weight_sets["hot"].items())
    x, _, _ = synth.make_batch(2, 32, 32, first_index=1400)
    assert np.array_equal(_run(src, x), _run(dst, x))
    assert np.array_equal(_run(dst, x), _run(dst, x))          # and the adopted blob is not repacked/invalidated


def test_native_library_is_loaded(models):
    """The parity above ran through libcid.so, not through any PyTorch op."""
    from celebrity_image_denoiser_amd import _lib

    maps = open("/proc/self/maps").read()
    assert os.path.realpath(_lib.LIB_PATH) in maps


@pytest.mark.parametrize("wset", ["default", "hot"])
def test_u8_in_u8_out_golden(models, golden_dir, wset):
    """uint8 HWC image -> uint8 HWC image with the reference's pre/post-processing folded
    into the first/last kernel, against the fixture made with the reference module (make_golden.py).
    The fp32 tensor must agree to 1e-5; the uint8 image is a truncation of it, so a value within 1e-5*255 of an
    integer may land one step away: allow |delta| <= 1 on at most 0.1 % of the bytes, 0 elsewhere."""
    import celebrity_image_denoiser_amd as cid

    g = np.load(os.path.join(golden_dir, f"u8_{wset}_32x40.npz"))
    m = models[wset]
    img = torch.from_numpy(g["noisy_u8"]).to("cuda:0")
    y_f32 = m.forward_u8(img, out_u8=False).cpu().numpy()
    assert np.abs(y_f32 - g["out_f32"]).max() <= TOL
    y_u8 = m.forward_u8(img).cpu().numpy()
    assert y_u8.shape == g["out_u8"].shape and y_u8.dtype == np.uint8
    d = np.abs(y_u8.astype(np.int16) - g["out_u8"].astype(np.int16))
    assert d.max() <= 1 and (d != 0).mean() <= 1e-3
    # host convenience wrapper: host uint8 in -> host uint8 out, identical bytes
    assert np.array_equal(cid.denoise_u8(m, torch.from_numpy(g["noisy_u8"])).numpy(), y_u8)
    # fused normalisation == explicit normalisation followed by the fp32 forward, bit for bit
    x = synth.normalize_u8(g["noisy_u8"])
    assert np.array_equal(_run(m, x), y_f32)


@pytest.mark.parametrize("wset", ["default", "hot"])
def test_serve_any_size_pad_and_crop(models, golden_dir, wset):
    """a 30x45 image through pad-to-multiple-of-4 -> network -> crop, like the reference server
    (app.py:276-281,384-385,474-480), against the fixture made with the reference module."""
    import celebrity_image_denoiser_amd as cid

    g = np.load(os.path.join(golden_dir, f"pad_{wset}_30x45.npz"))
    assert tuple(g["padding"]) == cid.get_padding(45, 30, 4) == (1, 1, 2, 1)
    y = cid.serve_u8(models[wset], torch.from_numpy(g["image_u8"])).numpy()
    assert y.shape == g["image_u8"].shape == g["out_u8"].shape
    d = np.abs(y.astype(np.int16) - g["out_u8"].astype(np.int16))
    assert d.max() <= 1 and (d != 0).mean() <= 1e-3


def test_padded_forward_equals_pad_in_memory_bit_for_bit(models):
    """f4 folded into the kernels (cid_forward_padded): the first kernel synthesises the black band, the last one writes only the
    crop window.  Property: for every caller-side format, every last-layer form and both storage types the result is, bit for
    bit, what padding the batch in memory (uint8 0 / fp32 -1.0), running the plain forward and slicing the window out gives —
    odd pads on all four sides, ragged tiles, a window that ends inside the last tile."""
    m = models["hot"]
    x, _, noisy = synth.make_batch(3, 37, 50, first_index=7100)
    xd, ud = torch.from_numpy(x).to("cuda:0"), torch.from_numpy(noisy).to("cuda:0")
    tails = ("tiles",) if m.conv_algo == "direct" else ("fused", "bands", "tiles")
    for pads in ((1, 1, 2, 1), (0, 3, 1, 0), (5, 2, 5, 5), (0, 0, 0, 0)):   # (left, top, right, bottom): 40x53, 40x51, 44x60 (37+7, 50+10), 37x50
        left, top, right, bottom = pads
        hp, wp = 37 + top + bottom, 50 + left + right
        fits = top + 37 <= 4 * (hp // 4) and left + 50 <= 4 * (wp // 4)
        for dtype in ("f32", "f16"):
            for tail in tails:
                m.compute_dtype, m.tail_algo = dtype, tail
                try:
                    if not fits:
                        with pytest.raises(RuntimeError, match="does not fit"):
                            m.forward_padded(ud, pads, out_u8=True)
                        continue
                    up = torch.nn.functional.pad(ud, (0, 0, left, right, top, bottom), value=0)
                    xp = torch.nn.functional.pad(xd, (left, right, top, bottom), value=-1.0)
                    want8 = m.forward_u8(up)[:, top:top + 37, left:left + 50, :]
                    wantf = m(xp)[:, :, top:top + 37, left:left + 50]
                    got8 = m.forward_padded(ud, pads, out_u8=True)
                    gotf = m.forward_padded(xd, pads, out_u8=False)
                    assert got8.shape == (3, 37, 50, 3) and gotf.shape == (3, 3, 37, 50)
                    assert torch.equal(got8, want8), (pads, dtype, tail)
                    assert torch.equal(gotf, wantf), (pads, dtype, tail)
                    assert torch.equal(m.forward_padded(xd, pads, out_u8=True), m.forward_fmt(xp, out_u8=True)[:, top:top + 37, left:left + 50, :])
                finally:
                    m.compute_dtype, m.tail_algo = "f32", _default_tail(m.conv_algo)


def test_walking_workgroups_equal_one_item_per_workgroup(weight_sets):
    """The Winograd F(4x2) launches WALK once they have more (tile, column block) items than two workgroups per CU: a workgroup
    takes tiles local, local + 64, ... of its XCD group with all column blocks of a tile back to back, the next tile's first chunk
    prefetched under the current tile's last one (wino42_kernels.h).  Property: the result is, bit for bit, that of one workgroup
    per item — on a RAGGED case: 37 images of 100x76 = 75 tiles each, so the XCD groups hold 347 tiles, not a multiple of the 64
    walkers, the last group is short, images straddle walkers and groups, and the quarter-resolution layers (13 tiles per image)
    fall back to one item per workgroup inside the same forward.  Plus a batch-independence and a CPU-oracle check of the walk."""
    _need_gpu()
    import celebrity_image_denoiser_amd as cid
    from celebrity_image_denoiser_amd import _lib
    from oracle import torch_oracle

    L = _lib.lib()
    m = cid.load(weight_sets["hot"], device="cuda:0", strict=True)
    x, _, _ = synth.make_batch(37, 100, 76, first_index=8100)
    xd = torch.from_numpy(x).to("cuda:0")
    prev = L.cid_debug_winograd_workgroups_per_cu(-1)
    assert prev == 2
    try:
        walk = m(xd).clone()
        assert L.cid_debug_winograd_workgroups_per_cu(0) == 2
        one = m(xd).clone()
        assert L.cid_debug_winograd_workgroups_per_cu(1) == 0          # one walker per CU: other strides, same bits
        single = m(xd).clone()
    finally:
        L.cid_debug_winograd_workgroups_per_cu(prev)
    assert torch.equal(walk, one) and torch.equal(walk, single)
    for i in (0, 17, 36):                                            # an image alone (no walking at all) = the same image in the batch
        assert torch.equal(m(xd[i:i + 1]), walk[i:i + 1]), i
    ref = torch_oracle.forward(weight_sets["hot"], x[[0, 36]]).numpy()
    assert np.abs(walk[[0, 36]].cpu().numpy() - ref).max() <= TOL


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_walking_full_size_every_stage_bit_equal_and_stable(weight_sets, dtype):
    """Full-size form of the two tests around it (64 images of 128x128: every walking launch takes several tiles per workgroup),
    stage by stage and repeated: every stored stage of the walking forward equals, bit for bit, that of the one-item-per-workgroup
    forward, three times in a row.  This is the comparison that located round 3's store-data hazard (pooled tensor, element 0 of
    four channel quads, run to run: profiles/r03_store_hazard.txt)."""
    _need_gpu()
    import celebrity_image_denoiser_amd as cid
    from celebrity_image_denoiser_amd import _lib

    L = _lib.lib()
    m = cid.load(weight_sets["hot"], device="cuda:0", strict=True)
    m.compute_dtype = dtype
    x, _, _ = synth.make_batch(64, 128, 128, first_index=8300)
    xd = torch.from_numpy(x).to("cuda:0")
    stages = ["down1", "pool1", "down2", "pool2", "bottleneck", "up2", "upconv2", "up1"]
    knob = L.cid_debug_winograd_workgroups_per_cu if dtype == "f32" else L.cid_debug_half_workgroups_per_cu
    prev = knob(-1)
    walkers = prev if dtype == "f32" else 3      # fp16: one item per workgroup is the default since round 4; walking (3 per CU) is the option under test
    assert walkers > 0

    def run(k):
        knob(k)
        y = m(xd).clone()
        torch.cuda.synchronize()
        return y, {s: m.stage_output(s, 64, 128, 128).clone() for s in stages}

    try:
        y0, s0 = run(0)
        for rep in range(3):
            y1, s1 = run(walkers)
            for s in stages:
                assert torch.equal(s0[s], s1[s]), (rep, s, int((s0[s] != s1[s]).sum()))
            assert torch.equal(y0, y1), rep
        if dtype == "f32":
            # round 4's measured option: one column block per XCD group on the launches with 2 / 4 column blocks (a.walk < 0) — another
            # walk order over the same items, so the same bits at every stage
            assert L.cid_debug_winograd_column_block_per_xcd(6) == 0
            try:
                y2, s2 = run(walkers)
            finally:
                assert L.cid_debug_winograd_column_block_per_xcd(0) == 6
            for s in stages:
                assert torch.equal(s0[s], s2[s]), ("xnb", s, int((s0[s] != s2[s]).sum()))
            assert torch.equal(y0, y2)
    finally:
        knob(prev)


def test_walking_workgroups_fp16_equal_one_item_per_workgroup(weight_sets):
    """The 3x3 launches of the fp16-storage path walk too (k_conv3x3_h16: next item's first B sub-chunk and halo chunk fetched under
    the last sub-step).  Same property on the same ragged case: three / one walker per CU and one workgroup per item give the same
    bits; plus the path's stated tolerance against the CPU oracle."""
    _need_gpu()
    import celebrity_image_denoiser_amd as cid
    from celebrity_image_denoiser_amd import _lib
    from oracle import torch_oracle

    L = _lib.lib()
    m = cid.load(weight_sets["hot"], device="cuda:0", strict=True)
    m.compute_dtype = "f16"
    x, _, _ = synth.make_batch(37, 100, 76, first_index=8100)
    xd = torch.from_numpy(x).to("cuda:0")
    prev = L.cid_debug_half_workgroups_per_cu(-1)
    assert prev == 0                                                  # the default since round 4: one item per workgroup
    try:
        one = m(xd).clone()
        assert L.cid_debug_half_workgroups_per_cu(3) == 0
        walk = m(xd).clone()
        assert L.cid_debug_half_workgroups_per_cu(1) == 3
        single = m(xd).clone()
    finally:
        L.cid_debug_half_workgroups_per_cu(prev)
    assert torch.equal(walk, one) and torch.equal(walk, single)
    for i in (0, 36):
        assert torch.equal(m(xd[i:i + 1]), walk[i:i + 1]), i
    ref = torch_oracle.forward(weight_sets["hot"], x[[0, 36]]).numpy()
    assert np.abs(walk[[0, 36]].cpu().numpy() - ref).max() <= 5e-3


def test_hip_graph_capture_and_replay(weight_sets):
    """cid_forward only enqueues kernels (no allocation, no synchronisation), so a forward can be captured into
    a HIP graph on the caller's stream and replayed: the replay must reproduce the eager result bit for bit and
    follow new input contents."""
    _need_gpu()
    import celebrity_image_denoiser_amd as cid

    m = cid.load(weight_sets["hot"], device="cuda:0", strict=True)
    x1 = torch.from_numpy(synth.make_batch(4, 64, 64, first_index=1500)[0]).to("cuda:0")
    x2 = torch.from_numpy(synth.make_batch(4, 64, 64, first_index=1600)[0]).to("cuda:0")
    static_x = x1.clone()
    m(static_x)                                     # packs weights and sizes the arena outside the capture
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s):
            static_y = m(static_x)
    torch.cuda.current_stream().wait_stream(s)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(static_y, m(x1))
    static_x.copy_(x2)
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(static_y, m(x2)) and not torch.equal(m(x1), m(x2))


@pytest.mark.parametrize("wset,tol", [("default", 5e-4), ("hot", 5e-3)])
def test_fp16_storage_path(weight_sets, golden_dir, wset, tol):
    """BASELINE configs[4]: half storage + fp16 MFMA (fp32 accumulators) against the fp32 golden output.
    A separate numerical contract: every stored activation and weight is rounded to half (2^-11 relative), so the
    stated tolerance is BASELINE.md section 4 / SURVEY 8(d)'s for this config: max|delta| <= 5e-3 — held on the He-gain
    ("hot") weights (activations up to 5, tanh to +-0.98; measured 2.7e-3, bench.py configs[4].max_abs_err_hot_weights)
    and with a factor 10 to spare at PyTorch-default weight scale (<= 5e-4; measured 4.4e-5).  PSNR delta vs the fp32
    reference is reported by the assertion message."""
    _need_gpu()
    import celebrity_image_denoiser_amd as cid
    from celebrity_image_denoiser_amd import psnr

    g = np.load(os.path.join(golden_dir, f"full_{wset}_128.npz"))
    x, clean, _ = synth.make_batch(2, 128, 128, 100)
    m = cid.load(weight_sets[wset], device="cuda:0", strict=True)
    m.compute_dtype = "f16"
    assert m.compute_dtype == "f16"
    y = _run(m, x)
    err = float(np.abs(y - g["out"]).max())
    dpsnr = abs(psnr(y, clean) - psnr(g["out"], clean))
    assert err <= tol, (err, dpsnr)
    assert dpsnr <= (0.01 if wset == "default" else 0.05), (err, dpsnr)
    # ragged / crop path and the u8 front/back ends work in half storage too
    gt = np.load(os.path.join(golden_dir, f"tiny_{wset}_13x18.npz"))
    assert np.abs(_run(m, gt["x"]) - gt["out"]).max() <= tol
    gu = np.load(os.path.join(golden_dir, f"u8_{wset}_32x40.npz"))
    yu = m.forward_u8(torch.from_numpy(gu["noisy_u8"]).to("cuda:0")).cpu().numpy().astype(np.int16)
    assert np.abs(yu - gu["out_u8"].astype(np.int16)).max() <= (1 if wset == "default" else 2)
    # the last layer's two forms on this path: fused (default since round 4: its 64 -> 9 x 4 contraction runs in upconv1[0]'s epilogue, z
    # rounded to half, k_conv_tail_zh sums the nine taps) and the separate tiled kernel on the stored 64-channel tensor
    from celebrity_image_denoiser_amd.generator import launch_table
    assert m.tail_algo == "fused"
    names = [r[1] for r in launch_table(2, 128, 128, m)]
    assert names[10].startswith("k_conv3x3_h16<128, 64, 0, true") and names[11].startswith("k_conv_tail_zh")
    m.tail_algo = "tiles"
    names = [r[1] for r in launch_table(2, 128, 128, m)]
    assert names[10].startswith("k_conv3x3_h16<128, 64, 0, false") and names[11].startswith("k_conv_tail_h")
    y_tiles = _run(m, x)
    assert float(np.abs(y_tiles - g["out"]).max()) <= tol
    assert float(np.abs(y_tiles - y).max()) <= (1e-4 if wset == "default" else 1.5e-3)     # what z's rounding to half costs (emulated: 2e-5 / 9e-4)
    assert np.abs(_run(m, gt["x"]) - gt["out"]).max() <= tol
    m.tail_algo = "fused"
    m.compute_dtype = "f32"
    assert np.abs(_run(m, x) - g["out"]).max() <= TOL


def test_fp16_ragged_shapes_and_long_walks(weight_sets):
    """The half-storage kernels away from 128x128: (a) ragged sizes against the ATen oracle at the fp16 tolerance — image areas
    that are not a multiple of the streaming transposed convolutions' 64 / 128-pixel runs (k_convt_t16 masks the run's tail per
    lane), widths that are not a multiple of 16 (an MFMA column tile then spans two image rows), cropped odd sizes; (b) a batch long
    enough that every persistent workgroup walks many runs (300 images: 9,600 runs of up1 for 512 workgroups, 4,800 of up2 for
    256), built from four distinct images: every copy must come out bit-identical to the first, and the first four within
    tolerance of the oracle (size-independent property: batch independence)."""
    _need_gpu()
    import celebrity_image_denoiser_amd as cid
    from oracle import torch_oracle

    m = cid.load(weight_sets["default"], device="cuda:0", strict=True)
    m.compute_dtype = "f16"
    for k, (n, h, w) in enumerate([(2, 52, 76), (3, 37, 150), (1, 100, 20), (2, 129, 67), (1, 8, 8)]):
        x, _, _ = synth.make_batch(n, h, w, first_index=5000 + 10 * k)
        y = _run(m, x)
        ref = torch_oracle.forward(weight_sets["default"], x).numpy()
        assert y.shape == ref.shape, (n, h, w)
        assert np.abs(y - ref).max() <= 5e-4, (n, h, w, float(np.abs(y - ref).max()))
    x4, _, _ = synth.make_batch(4, 128, 128, first_index=5100)
    x = np.ascontiguousarray(np.tile(x4, (75, 1, 1, 1)))
    y = _run(m, x)
    ref = torch_oracle.forward(weight_sets["default"], x4).numpy()
    assert np.abs(y[:4] - ref).max() <= 5e-4
    y = y.reshape(75, 4, *y.shape[1:])
    assert np.array_equal(y, np.broadcast_to(y[:1], y.shape))


def test_host_pipeline_matches_direct_calls(models):
    """HostPipeline (upload / forward / download on three streams over two slots) returns, in order, exactly the
    bytes the plain calls return: uint8 and fp32 batches, a ragged last batch, a change of image size mid-stream,
    iterated denoising, and the pinned-view mode."""
    from celebrity_image_denoiser_amd import HostPipeline

    m = models["default"]
    rng = np.random.default_rng(7)
    u8 = [rng.integers(0, 256, size=(n, 24, 36, 3), dtype=np.uint8) for n in (5, 5, 5, 5, 3)]
    u8 += [rng.integers(0, 256, size=(2, 17, 20, 3), dtype=np.uint8)]           # new image size: slots are refitted
    want = [m.forward_u8(torch.from_numpy(b).to("cuda:0")).cpu().numpy() for b in u8]
    pipe = HostPipeline(m, depth=2)
    got = [t.numpy() for t in pipe.run(u8)]
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g.shape == w.shape and g.dtype == np.uint8 and np.array_equal(g, w)
    # pinned views (copy=False) are valid until the generator advances
    for t, w in zip(pipe.run(u8[:4], copy=False), want[:4]):
        assert t.is_pinned() and np.array_equal(t.numpy(), w)
    # fp32 batches, three slots, two iterations == forward(forward(x))
    f32 = [synth.make_batch(n, 20, 24, first_index=100 + 10 * i)[0] for i, n in enumerate((3, 3, 2))]
    want2 = [_run(m, _run(m, b)) for b in f32]
    got2 = [t.numpy() for t in HostPipeline(m, depth=3).run(f32, iterations=2)]
    for g, w in zip(got2, want2):
        assert np.array_equal(g, w)
    # uint8 with iterations: the second pass consumes the fp32 output of the first (denoise_eavl_iter.py:93-96)
    z = m.forward_u8(torch.from_numpy(u8[0]).to("cuda:0"), out_u8=False)
    want3 = m.forward_fmt(z, out_u8=True).cpu().numpy()
    assert np.array_equal(next(iter(pipe.run(u8[:1], iterations=2))).numpy(), want3)
    with pytest.raises(RuntimeError):
        list(pipe.run([np.zeros((1, 3, 3, 3), np.uint8)]))                      # H < 4: loud, like the reference
    with pytest.raises(RuntimeError):
        list(pipe.run([np.zeros((1, 8, 8), np.float32)]))


def test_out_argument_and_mixed_formats(models):
    """forward(x, out=) / forward_fmt(x, out_u8, out=) write into caller-owned tensors and refuse wrong ones."""
    m = models["hot"]
    x, _, noisy = synth.make_batch(2, 16, 20, first_index=40)
    xd = torch.from_numpy(x).to("cuda:0")
    y = m(xd)
    buf = torch.empty_like(y)
    assert m(xd, out=buf) is buf and torch.equal(buf, y)
    with pytest.raises(RuntimeError):
        m(xd, out=torch.empty((2, 3, 16, 24), device="cuda:0"))
    with pytest.raises(RuntimeError):
        m(xd, out=torch.empty((2, 3, 16, 20), dtype=torch.float16, device="cuda:0"))
    # fp32 in -> uint8 out equals uint8 in -> uint8 out when the fp32 input is the normalised uint8 image
    img = torch.from_numpy(noisy).to("cuda:0")
    a = m.forward_u8(img)
    b = m.forward_fmt(torch.from_numpy(synth.normalize_u8(noisy)).to("cuda:0"), out_u8=True)
    assert a.dtype == torch.uint8 and torch.equal(a, b)


def test_seeded_shape_sweep_against_cpu_oracle(models, weight_sets):
    """Fourteen seeded (N, H, W) draws with H, W in [4, 150] — tile boundaries of every kernel (32/64-pixel tile columns,
    2/4/8-row tiles, the W <= 32 variants), odd sizes that exercise the crop path, single rows of tiles — against the
    ATen oracle at the stated tolerance."""
    from oracle import torch_oracle

    rng = np.random.default_rng(20240607)
    shapes = [(int(rng.integers(1, 4)), int(rng.integers(4, 151)), int(rng.integers(4, 151))) for _ in range(10)]
    shapes += [(1, 33, 65), (2, 130, 31), (1, 20, 36), (2, 68, 132)]   # the last two: every level a partial 32x4 / 16x8 Winograd F(4x2) workgroup tile
    for k, (n, h, w) in enumerate(shapes):
        wset = "hot" if k % 2 else "default"
        x, _, _ = synth.make_batch(n, h, w, first_index=2000 + 10 * k)
        y = _run(models[wset], x)
        ref = torch_oracle.forward(weight_sets[wset], x).numpy()
        assert y.shape == ref.shape == (n, 3, 4 * (h // 4), 4 * (w // 4)), (n, h, w)
        assert np.abs(y - ref).max() <= TOL, (n, h, w, float(np.abs(y - ref).max()))


def test_full_size_translation_of_batch_order(models):
    """Size-independent property at BASELINE config-2 size: permuting the images of a batch permutes the outputs,
    bit for bit (tile -> workgroup assignment, XCD placement and tiles-per-workgroup grouping must not leak between images)."""
    x, _, _ = synth.make_batch(40, 128, 128, first_index=3000)
    perm = np.random.default_rng(5).permutation(40)
    y = _run(models["default"], x)
    yp = _run(models["default"], np.ascontiguousarray(x[perm]))
    assert np.array_equal(yp, y[perm])


def test_graphed_forward_matches_eager(models):
    """GraphedForward (one HIP-graph launch per forward) returns the bits of the eager call, for fp32 and uint8 inputs,
    follows new input contents, and refuses other shapes loudly."""
    from celebrity_image_denoiser_amd import GraphedForward

    m = models["default"]
    x, _, noisy = synth.make_batch(3, 24, 40, first_index=4000)
    xd = [torch.from_numpy(x[i:i + 1]).to("cuda:0") for i in range(3)]
    fast = GraphedForward(m, xd[0])
    for t in xd:
        assert torch.equal(fast(t), m(t))
    ud = [torch.from_numpy(noisy[i:i + 1]).to("cuda:0") for i in range(3)]
    fast8 = GraphedForward(m, ud[0])
    for t in ud:
        assert torch.equal(fast8(t), m.forward_u8(t))
    with pytest.raises(RuntimeError):
        fast(torch.zeros((2, 3, 24, 40), device="cuda:0"))


def test_graphed_forward_survives_arena_growth_and_new_weights(weight_sets):
    """the captured graph holds raw pointers to the arena and the packed blob.  A later eager call with a
    bigger batch replaces the model's arena, a parameter update its blob: the graph must keep working (private arena)
    and follow the new weights (re-capture), not replay into freed memory or with stale weights."""
    _need_gpu()
    import celebrity_image_denoiser_amd as cid
    from celebrity_image_denoiser_amd import GraphedForward

    m = cid.load(weight_sets["default"], device="cuda:0", strict=True)
    x1 = torch.from_numpy(synth.make_batch(1, 24, 40, first_index=4100)[0]).to("cuda:0")
    fast = GraphedForward(m, x1)
    want = m(x1).clone()
    big = torch.from_numpy(synth.make_batch(16, 96, 96, first_index=4200)[0]).to("cuda:0")
    m(big)                                                   # grows (replaces) the model's own arena
    junk = [torch.full((1 << 20,), float("nan"), device="cuda:0") for _ in range(8)]   # reuse whatever was freed
    torch.cuda.synchronize()
    assert torch.equal(fast(x1), want)
    del junk
    m.load_state_dict({k: torch.from_numpy(v) for k, v in weight_sets["hot"].items()}, strict=True)
    want_hot = m(x1).clone()
    assert not torch.equal(want_hot, want)
    assert torch.equal(fast(x1), want_hot)                   # noticed the new blob and re-captured


def test_bench_two_ranks_sharing_the_gpu():
    """The N > 1 path of bench.py with REAL GPU processes on a one-GPU box (`--rehearse-shared-gpu`: both ranks on cuda:0, gloo): the plain
    command starts its own ranks, they rendezvous, RCCL's set-up is refused on both (two ranks on one device) and the agreed fallback transport
    carries the blob, each rank runs its shard, and rank 0's single line reports the aggregate and the worst oracle error over BOTH ranks —
    rank 1 started from its own random init, so it passes only on the broadcast weights (backend/app.py:80-103 is the shard unit)."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-shared-gpu", "--batch-per-gpu", "8", "--steps", "2", "--warmup", "1",
                          "--no-extras", "--no-cpu-baseline"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, out.stdout[-2000:]                            # ONE line on stdout, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 16 and "rehearsal" in d and d["scaling"] == "weak"
    assert d["rccl"]["collectives_in_forward"] == 0 and d["rccl"]["transport"] in ("torch-distributed", "rccl-cabi")
    assert d["parity"]["max_abs_err_all_ranks"] <= 1e-5 and d["parity"]["max_abs_err_vs_cpu_oracle"] <= 1e-5
    assert d["value"] == pytest.approx(16 * 2 / (d["ms_per_step"] * 2 * 1e-3), rel=1e-3)     # whole-job images over the max-over-ranks time


def test_rccl_broadcast_through_the_c_abi(weight_sets):
    """the one collective of the multi-GPU job is `cid_broadcast_weights` — ncclBroadcast issued by
    libcid.so on a communicator made with cid_comm_* — here on the world this box has (one rank).  The receiver's
    half (refresh of the host copy from the broadcast device blob) is driven by passing rank != root to the handle
    of a second module attached to a copy of the blob."""
    _need_gpu()
    import ctypes

    import torch.distributed as dist

    import celebrity_image_denoiser_amd as cid
    from celebrity_image_denoiser_amd import _lib
    from celebrity_image_denoiser_amd import dist as cdist

    if not dist.is_initialized():
        dist.init_process_group("gloo", init_method="tcp://127.0.0.1:29577", rank=0, world_size=1)
    L_available = lambda: _lib.lib().cid_comm_available() == 1   # noqa: E731
    try:
        src = cid.load(weight_sets["hot"], device="cuda:0", strict=True)
        x, _, _ = synth.make_batch(2, 32, 32, first_index=1400)
        want = _run(src, x)
        first = cdist.broadcast_weights_ex(src, src=0)      # communicator set-up + ncclBroadcast (root side); the communicator is kept
        assert first["transport"] == "rccl-cabi" and first["nranks"] == 1 and first["setup_ms"] > 0.0   # nranks: what RCCL itself reports
        again = cdist.broadcast_weights_ex(src, src=0)      # second broadcast: the cached communicator, only the transfer
        assert again["transport"] == "rccl-cabi" and again["setup_ms"] == 0.0 and again["broadcast_ms"] < first["setup_ms"]
        assert np.array_equal(_run(src, x), want)
        assert L_available()
        comm = cdist.WeightsComm.negotiate(torch.device("cuda:0"))
        assert comm is not None
        dst = cid.load(None, device="cuda:0")               # random init, like a rank that read no checkpoint
        blob = src.pack_weights().clone()
        L = _lib.lib()
        _lib.check(dst._cid, L.cid_attach_weights(dst._cid, blob.data_ptr()))
        stream = torch.cuda.current_stream().cuda_stream
        _lib.check(dst._cid, L.cid_broadcast_weights(dst._cid, comm._comm, 0, 1, stream))   # receiver role
        dst.adopt_packed_weights(blob, update_parameters=True, host_is_current=True)
        comm.close()
        assert all(torch.equal(dst.state_dict()[k].cpu(), torch.from_numpy(v)) for k, v in weight_sets["hot"].items())
        assert np.array_equal(_run(dst, x), want)
        h = ctypes.c_void_p()
        L.cid_create(ctypes.byref(h))
        assert L.cid_broadcast_weights(h, ctypes.c_void_p(8), 0, 0, None) == 4 and b"no device blob" in L.cid_last_error(h)
        L.cid_destroy(h)
    finally:
        cdist.WeightsComm.close_all()
        dist.destroy_process_group()


def test_striped_forward_equals_single_call_bit_for_bit(models):
    """Images beyond one call's size limit (cid_forward refuses H*W >= 4,194,303: 32-bit per-image addressing) are cut into
    horizontal stripes with a 32-row halo (generator._forward_striped).  Property: with stripes FORCED on a mid-size image
    the assembled result is the single-call result, bit for bit — fp32 and uint8 formats, a height that is not a multiple
    of 4 (crop path at the bottom), stripes of 16, 32 and 64 rows (multiples of 16: the 4x4 Winograd tiles keep their alignment at quarter resolution)."""
    m = models["hot"]
    x, _, noisy = synth.make_batch(2, 150, 70, first_index=6000)
    xd, ud = torch.from_numpy(x).to("cuda:0"), torch.from_numpy(noisy).to("cuda:0")
    want, want8 = m(xd).clone(), m.forward_u8(ud).clone()
    for rows in (16, 32, 64):
        assert torch.equal(m._forward_striped(xd, out_u8=False, stripe_rows=rows), want), (rows, float((m._forward_striped(xd, out_u8=False, stripe_rows=rows) - want).abs().max()))
        assert torch.equal(m._forward_striped(ud, out_u8=True, stripe_rows=rows), want8), rows
    with pytest.raises(RuntimeError):
        m._forward_striped(xd, out_u8=False, stripe_rows=24)


def test_image_beyond_the_single_call_limit(models, weight_sets):
    """2048x2080 (the ADVICE r1 case: accepted in round 1, with real activations read as zero padding): now striped
    automatically.  Checked against the CPU oracle on two windows cut around a stripe seam and at the bottom edge — a window
    with 40 rows of context reproduces the full image's values there (receptive field +-20 rows)."""
    from oracle import torch_oracle

    free, _ = torch.cuda.mem_get_info()
    if free < 60 * 2**30:
        pytest.skip("needs ~40 GiB of free device memory")
    m = models["default"]
    h, w = 2080, 2048
    assert m._needs_stripes(h, w)
    rng = np.random.default_rng(11)
    x = (rng.random((1, 3, h, w), dtype=np.float32) * 2 - 1)
    y = m(torch.from_numpy(x).to("cuda:0")).cpu().numpy()
    assert y.shape == (1, 3, h, w) and np.isfinite(y).all()
    rows_per = (m.MAX_PIXELS_PER_CALL // w - 2 * m.STRIPE_HALO) // 16 * 16   # the product's stripe height (generator._forward_striped)
    assert 0 < rows_per < h
    for r0 in (rows_per - 8, h - 16):          # a window with 8 rows on either side of the first seam; the bottom edge
        lo, hi = max(0, r0 - 40) // 4 * 4, min(h, r0 + 16 + 40)
        ref = torch_oracle.forward(weight_sets["default"], x[:, :, lo:hi, 512:768 + 64]).numpy()
        got = y[:, :, r0:r0 + 16, 512 + 32:768 + 32]
        assert np.abs(got - ref[:, :, r0 - lo:r0 - lo + 16, 32:-32]).max() <= TOL, r0
    m._ws = None
    torch.cuda.empty_cache()


def test_batch_past_2_31_elements_per_buffer(models):
    """Maximum-size edge: 2080 images of 128x128 make the 64-channel buffers hold more than 2^31 elements (60 GiB arena).
    Every 16-image group must equal the same 16 images run alone, bit for bit (64-bit base addresses, 32-bit per-image
    offsets, tile decode by multiply-high)."""
    free, _ = torch.cuda.mem_get_info()
    if free < 90 * 2**30:
        pytest.skip("needs ~70 GiB of free device memory")
    m = models["hot"]
    n = 2080
    base, _, _ = synth.make_batch(16, 128, 128, first_index=7000)
    x = torch.from_numpy(base).to("cuda:0").repeat(n // 16, 1, 1, 1).contiguous()
    assert n * 128 * 128 * 64 > 2**31
    y = m(x)
    ref = m(x[:16].contiguous())
    torch.cuda.synchronize()
    assert all(torch.equal(y[k:k + 16], ref) for k in range(0, n, 16))
    del x, y
    m._ws = None            # give the arena back
    torch.cuda.empty_cache()


@pytest.mark.parametrize("dtype", ["f32", "f16"])
def test_no_kernel_reads_unwritten_lds_or_arena(models, weight_sets, dtype):
    """Every byte of the CUs' LDS and of the activation arena is NaN before the forward (cid_debug_poison_lds; LDS is not
    cleared between kernels).  A kernel that reads an LDS word or an arena element nobody has written would put NaN (or,
    after a ReLU, a wrong 0) into the result: the output must be the bits of an unpoisoned run.  Multi-tile, ragged and
    minimum shapes, uint8 and fp32 callers' formats."""
    from celebrity_image_denoiser_amd import _lib

    m = models["hot"]
    m.compute_dtype = dtype
    try:
        stream = torch.cuda.current_stream().cuda_stream
        for (n, h, w) in ((3, 128, 128), (2, 40, 72), (1, 21, 30), (2, 4, 4)):
            x, _, noisy = synth.make_batch(n, h, w, first_index=5000)
            xd, ud = torch.from_numpy(x).to("cuda:0"), torch.from_numpy(noisy).to("cuda:0")
            want, want8 = m(xd).clone(), m.forward_u8(ud).clone()
            for fn, ref in ((lambda: m(xd), want), (lambda: m.forward_u8(ud), want8)):
                m._ws.view(torch.float32).fill_(float("nan"))
                assert _lib.lib().cid_debug_poison_lds(stream) == 0
                got = fn()
                torch.cuda.synchronize()
                assert torch.equal(got, ref), (dtype, n, h, w)
    finally:
        m.compute_dtype = "f32"


@pytest.mark.parametrize("wset", ["default", "hot"])
def test_white_noise_inputs_stay_within_the_fp32_contract(models, weight_sets, wset):
    """VERDICT r3 #4(i).  The reference accepts ANY image in [-1, 1] (app.py:400-406), not only the smooth face-like fields the
    benchmark batch is made of; white noise is the worst case for the Winograd transforms' rounding (no cancellation between
    neighbouring pixels).  Uniform noise in [-1, 1] at a ragged width (128 x 134: partial tiles at the right edge), every
    algorithm (module fixture) x both weight sets against the ATen oracle at the stated 1e-5.  Measured in round 3 with the
    ad-hoc tools/noise_err.py: 4e-6 ... 7e-6 on the He-gain set (the 9-tap direct kernel the largest), 3e-8 at default scale."""
    from oracle import torch_oracle

    g = torch.Generator(device="cpu")
    g.manual_seed(99)
    x = (torch.rand((6, 3, 128, 134), generator=g) * 2 - 1).contiguous().numpy()
    ref = torch_oracle.forward(weight_sets[wset], x).numpy()
    y = _run(models[wset], x)
    assert y.shape == ref.shape == (6, 3, 128, 132)
    err = float(np.abs(y - ref).max())
    assert err <= TOL, (models[wset].conv_algo, wset, err)
    # the extreme members of the contract: a saturated checkerboard (every pixel +-1, sign alternating per pixel and channel)
    yy, xx = np.mgrid[0:64, 0:64]
    cb = np.stack([((yy + xx + c) % 2) * 2.0 - 1.0 for c in range(3)]).astype(np.float32)[None]
    err_cb = float(np.abs(_run(models[wset], cb) - torch_oracle.forward(weight_sets[wset], cb).numpy()).max())
    assert err_cb <= TOL, (models[wset].conv_algo, wset, err_cb)


def test_split16_wide_images_batch_independence_and_error_budget(weight_sets):
    """conv_algo="split16" (opt-in; include/cid.h CID_ALGO_SPLIT16) beyond what the per-algorithm fixture sees: (a) 256 x 256 images (the row-band last layer
    does not take them: the handle falls back to the tiled one) and a ragged 250 x 300 against the ATen oracle at the fp32 path's 1e-5, on the He-gain weights;
    (b) batch independence, bit for bit: images of a 40-image batch (grid of 10,240 workgroups per layer) equal the same images run alone; (c) on 8 He-gain images
    the split form and the exact-fp32 direct kernel stand at the same distance from the ATen oracle (measured 5.8e-6 and 5-7e-6: fp32-grade sums in different orders), both inside 1e-5."""
    _need_gpu()
    import celebrity_image_denoiser_amd as cid
    from oracle import torch_oracle

    sd = weight_sets["hot"]
    m = cid.load(sd, device="cuda:0", strict=True)
    m.conv_algo, m.tail_algo = "split16", "bands"
    for shape in ((2, 256, 256), (1, 250, 300)):
        x, _, _ = synth.make_batch(shape[0], shape[1], shape[2], first_index=4242)
        y = _run(m, x)
        ref = torch_oracle.forward(sd, x).numpy()
        assert y.shape == ref.shape and float(np.abs(y - ref).max()) <= TOL, shape
    x, _, _ = synth.make_batch(40, 128, 128, first_index=5100)
    xd = torch.from_numpy(x).to("cuda:0")
    yb = m(xd).clone()
    for i in (0, 19, 39):
        assert torch.equal(m(xd[i:i + 1].contiguous()), yb[i:i + 1]), i
    ref8 = torch_oracle.forward(sd, x[:8]).numpy()
    err = float(np.abs(yb[:8].cpu().numpy() - ref8).max())
    md = cid.load(sd, device="cuda:0", strict=True)
    md.conv_algo, md.tail_algo = "direct", "bands"
    err_direct = float(np.abs(_run(md, x[:8]) - ref8).max())
    assert err <= TOL and err_direct <= TOL and err <= 2.0 * err_direct, (err, err_direct)
    from celebrity_image_denoiser_amd import _lib

    names = [_lib.lib().cid_launch_kernel(m._cid, i).decode() for i in range(12)]
    # the eight 3x3 layers AND the two transposed convolutions run on the split-operand kernel; head and last layer are the fp32 path's own
    assert sum(n.startswith("k_conv3x3_h16<") and n.endswith(", false, false, true,") for n in names) == 10 and names[0].startswith("k_conv_head")
    assert names[6] == "k_conv3x3_h16<256, 128, 2, false, false, true," and names[9] == "k_conv3x3_h16<128, 64, 2, false, false, true,"
    # the fused last layer under this algorithm (the default form): upconv1[2]'s contraction in split-operand arithmetic inside upconv1[0]'s kernel, 27 fp32 planes to k_conv_tail_z;
    # same contract, and the launch table says so
    m.tail_algo = "fused"
    yf = m(xd)
    assert float(np.abs(yf[:8].cpu().numpy() - ref8).max()) <= TOL and float((yf - yb).abs().max()) <= TOL
    assert _lib.lib().cid_launch_kernel(m._cid, 10).decode() == "k_conv3x3_h16<128, 64, 0, true, false, true," and _lib.lib().cid_launch_kernel(m._cid, 11).decode().startswith("k_conv_tail_z<")
    for shape in ((1, 250, 300),):
        x2, _, _ = synth.make_batch(shape[0], shape[1], shape[2], first_index=4243)
        assert float(np.abs(_run(m, x2) - torch_oracle.forward(sd, x2).numpy()).max()) <= TOL


def test_config3_full_size_256x256_batch_256(weight_sets):
    """BASELINE configs[3] at its REAL size — B = 256 images of 256 x 256 (31.7 GB arena) — which the suite so far
    saw only at N = 1.  Size-independent properties, on the default algorithm: (a) batch independence: three images of the
    full batch equal, bit for bit, the same images run alone (different grids: the full batch walks, a single image does not);
    (b) one image against the CPU oracle at 1e-5; (c) every stored stage of the walking launch equals the one-item-per-workgroup
    launch bit for bit."""
    _need_gpu()
    import celebrity_image_denoiser_amd as cid
    from celebrity_image_denoiser_amd import _lib
    from oracle import torch_oracle

    free, _ = torch.cuda.mem_get_info()
    if free < 80 * 2**30:
        pytest.skip("needs ~70 GiB of free device memory")
    L = _lib.lib()
    B, S = 256, 256
    m = cid.load(weight_sets["hot"], device="cuda:0", strict=True)
    base, _, _ = synth.make_batch(8, S, S, first_index=9100)                # eight distinct images, tiled to the full batch
    xd = torch.from_numpy(base).to("cuda:0").repeat(B // 8, 1, 1, 1).contiguous()
    stages = ["down1", "pool1", "down2", "pool2", "bottleneck", "up2", "upconv2", "up1"]
    prev = L.cid_debug_winograd_workgroups_per_cu(-1)
    try:
        y = m(xd)
        torch.cuda.synchronize()
        probe = (0, 131, 255)
        walk_stage = {s: m.stage_output(s, B, S, S)[list(probe)].clone() for s in stages}
        for i in probe:                                                     # (a)
            assert torch.equal(m(xd[i:i + 1].contiguous()), y[i:i + 1]), i
        for k in range(8, B, 8):                                            # copies of the same eight images: identical bits
            assert torch.equal(y[k:k + 8], y[:8]), k
        ref = torch_oracle.forward(weight_sets["hot"], base[:1]).numpy()    # (b)
        err = float(np.abs(y[:1].cpu().numpy() - ref).max())
        assert err <= TOL, err
        L.cid_debug_winograd_workgroups_per_cu(0)                           # (c)
        y_one = m(xd)
        torch.cuda.synchronize()
        assert torch.equal(y_one, y)
        for s in stages:
            one = m.stage_output(s, B, S, S)[list(probe)]
            assert torch.equal(one, walk_stage[s]), (s, int((one != walk_stage[s]).sum()))
    finally:
        L.cid_debug_winograd_workgroups_per_cu(prev)
        del xd
        m._ws = None
        torch.cuda.empty_cache()


def test_fp16_storage_contract_on_he_gain_weights_16_images(weight_sets):
    """BASELINE.md section 4 states max|delta| <= 5e-3 for configs[4] (fp16 storage, fp16 MFMA, fp32 accumulate).
    Until now the 16-image He-gain measurement lived only in bench.py's configs leg (3.0e-3); here it is a test: 16 images of
    128 x 128 from the benchmark's generator, He-gain ("hot") weights (activations up to ~5, tanh to +-0.98), against the fp32
    ATen oracle; and the PSNR of the result against the clean images within 0.05 dB of the fp32 reference's."""
    _need_gpu()
    import celebrity_image_denoiser_amd as cid
    from celebrity_image_denoiser_amd import psnr
    from oracle import torch_oracle

    m = cid.load(weight_sets["hot"], device="cuda:0", strict=True)
    m.compute_dtype = "f16"
    x, clean, _ = synth.make_batch(16, 128, 128, first_index=100)
    y = _run(m, x)
    ref = torch_oracle.forward(weight_sets["hot"], x).numpy()
    err = float(np.abs(y - ref).max())
    assert err <= 5e-3, err
    assert abs(psnr(y, clean) - psnr(ref, clean)) <= 0.05
    # and at full batch (B = 512, configs[4]'s size): the first 16 images of the big batch are the bits of the 16 run alone
    xb = torch.from_numpy(x).to("cuda:0").repeat(32, 1, 1, 1).contiguous()
    yb = m(xb)
    torch.cuda.synchronize()
    assert torch.equal(yb[:16].cpu(), torch.from_numpy(y)) and torch.equal(yb[496:], yb[:16])
    del xb, yb
    m._ws = None
    torch.cuda.empty_cache()
