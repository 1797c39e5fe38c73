"""Host-side mirror of the reference interface, without a GPU: construction, state_dict names and
shapes, checkpoint unwrapping (load_state_safely semantics), strictness, loud failure without a GPU,
synthetic data recipe, PSNR definition, shard arithmetic."""
import os

import numpy as np
import pytest
import torch

import celebrity_image_denoiser_amd as cid

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from celebrity_image_denoiser_amd import api, dist as cdist, synth


def test_module_surface_matches_reference_keys():
    m = cid.DenoiseGenerator()
    sd = m.state_dict()
    assert list(sd.keys()) == list(synth.param_shapes().keys())
    assert {k: tuple(v.shape) for k, v in sd.items()} == dict(synth.param_shapes())
    assert sum(v.numel() for v in sd.values()) == 1827587
    assert m.eval() is m
    assert isinstance(m, torch.nn.Module)


def test_checkpoint_unwrapping_like_load_state_safely(tmp_path):
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict("hot").items()}
    # (i) trainer layout {"generator": sd, ...} (training.py:359-376), (ii) bare, (iii) "module."-prefixed
    for ckpt in ({"generator": sd, "epoch": 3}, sd, {"state_dict": {"module." + k: v for k, v in sd.items()}}, {"G": sd}):
        got = api.extract_state_dict(ckpt)
        assert list(got.keys()) == list(sd.keys())
    path = os.path.join(tmp_path, "denoise_epoch_499.pth")
    torch.save({"generator": {"module." + k: v for k, v in sd.items()}, "epoch": 499}, path)
    m = cid.DenoiseGenerator()
    api.load_state_safely(m, path)
    assert all(torch.equal(m.state_dict()[k], sd[k]) for k in sd)
    assert not m.training


def test_strict_and_non_strict_loading():
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict("default").items()}
    m = cid.DenoiseGenerator()
    partial = {k: v for k, v in sd.items() if not k.startswith("up1")}
    with pytest.raises(RuntimeError):
        m.load_state_dict(partial, strict=True)
    res = m.load_state_dict(partial, strict=False)                   # app.py:272 uses strict=False
    assert sorted(res.missing_keys) == ["up1.bias", "up1.weight"]
    with pytest.raises(RuntimeError):                                # wrong shape is an error even non-strict
        m.load_state_dict({"up2.weight": torch.zeros(128, 256, 2, 2)}, strict=False)


def test_no_cpu_fallback():
    m = cid.DenoiseGenerator()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 8, 8))
    with pytest.raises(RuntimeError):
        m.pack_weights()
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="GPU"):
            cid.load(None)


def test_missing_library_fails_loudly(monkeypatch):
    from celebrity_image_denoiser_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libcid.so")
    with pytest.raises(RuntimeError, match="not built"):
        _lib.lib()


def test_host_packed_blob_roundtrip_refreshes_parameters():
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict("hot").items()}
    a, b = cid.DenoiseGenerator(), cid.DenoiseGenerator()
    a.load_state_dict(sd)
    b.adopt_packed_weights(a.pack_weights_host())
    assert all(torch.equal(b.state_dict()[k], sd[k]) for k in sd)
    with pytest.raises(ValueError):
        b.adopt_packed_weights(torch.zeros(16, dtype=torch.uint8))


def test_synthetic_data_recipe():
    x, clean, noisy = synth.make_batch(2, 32, 48, first_index=5)
    assert x.shape == (2, 3, 32, 48) and x.dtype == np.float32 and noisy.dtype == np.uint8
    assert x.min() >= -1 and x.max() <= 1
    # x = (u8/255 - 0.5)/0.5 in float32 (app.py:401-405)
    assert np.array_equal(x, ((noisy.astype(np.float32) / np.float32(255) - np.float32(0.5)) / np.float32(0.5)).transpose(0, 3, 1, 2))
    x2, _, _ = synth.make_batch(1, 32, 48, first_index=6)
    assert np.array_equal(x2[0], x[1])                               # image index, not batch position, seeds an image
    d = noisy.astype(np.float64) - synth.clean_images_u8(2, 32, 48, 5)
    assert 15 < d.std() < 30                                         # sigma = 25 before clipping (noise_generation.py:6-10)
    a, b = synth.make_state_dict("default"), synth.make_state_dict("hot")
    assert abs(a["down2.0.weight"]).max() <= 1 / np.sqrt(64 * 9) and abs(b["down2.0.weight"]).max() > 1 / np.sqrt(64 * 9)


def test_psnr_definition():
    rng = np.random.default_rng(0)
    a = rng.uniform(-1, 1, (3, 3, 8, 8)).astype(np.float32)
    b = a + rng.normal(0, 0.1, a.shape).astype(np.float32)
    mse = ((a.astype(np.float64) - b) ** 2).reshape(3, -1).mean(1)
    assert abs(cid.psnr(a, b) - np.mean(10 * np.log10(4.0 / mse))) < 1e-12   # data_range=2.0, training.py:382
    assert cid.psnr(torch.from_numpy(a), torch.from_numpy(a)) == float("inf")


def test_shard_range_partitions_the_batch():
    for n, w in ((2048, 8), (256, 1), (10, 4), (3, 8), (0, 2)):
        spans = [cdist.shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        assert max(e - b for b, e in spans) - min(e - b for b, e in spans) <= 1
    assert cdist.shard_range(2048, 3, 8) == (768, 1024)              # config 3: 256 images per GPU
    with pytest.raises(ValueError):
        cdist.shard_range(8, 2, 2)


def test_get_padding_rule():
    """app.py:276-281: pad to a multiple of divisor*scale, extra pixel on the right/bottom."""
    assert cid.get_padding(128, 128) == (0, 0, 0, 0)
    assert cid.get_padding(45, 30) == (1, 1, 2, 1)
    assert cid.get_padding(127, 130) == (0, 1, 1, 1)
    assert cid.get_padding(5, 5, divisor=4, scale=4) == (5, 5, 6, 6)


def test_torch_free_checkpoint_reader(tmp_path):
    """read the trainer's checkpoint layout (training.py:359-376) without torch.load."""
    from celebrity_image_denoiser_amd import ckpt

    sd = {("module." + k): torch.from_numpy(v) for k, v in synth.make_state_dict("hot").items()}
    # a non-contiguous tensor and a 0-dim tensor exercise stride / scalar handling
    extra = torch.arange(24, dtype=torch.float32).reshape(4, 6).t()
    path = os.path.join(tmp_path, "denoise_epoch_499.pth")
    torch.save({"generator": sd, "discriminator": {"w": extra, "s": torch.tensor(3.5)}, "epoch": 499,
                "best_psnr": 31.25, "metric_history": {"psnr": [1.0, 2.0]},
                "g_optimizer": {"state": {}, "param_groups": [{"lr": 1e-4, "betas": (0.9, 0.999), "params": [0, 1]}]}}, path)
    whole = ckpt.read_checkpoint(path)
    assert whole["epoch"] == 499 and whole["best_psnr"] == 31.25 and whole["g_optimizer"]["param_groups"][0]["betas"] == (0.9, 0.999)
    assert np.array_equal(whole["discriminator"]["w"], extra.numpy()) and float(whole["discriminator"]["s"]) == 3.5
    got = ckpt.read_state_dict(path)
    ref = synth.make_state_dict("hot")
    assert list(got.keys()) == list(ref.keys())
    assert all(np.array_equal(got[k], ref[k]) and got[k].dtype == np.float32 for k in ref)
    # and it feeds the module like torch.load would
    m = cid.DenoiseGenerator()
    m.load_state_dict({k: torch.from_numpy(v) for k, v in got.items()}, strict=True)
    # a pickle that names anything outside the allow-list is refused, not executed
    import pickle, zipfile
    evil = os.path.join(tmp_path, "evil.pth")
    with zipfile.ZipFile(evil, "w") as zf:
        zf.writestr("archive/data.pkl", pickle.dumps(os.getcwd))
    with pytest.raises(pickle.UnpicklingError):
        ckpt.read_checkpoint(evil)
    # ... and the path-taking loader does NOT retry it with a more permissive unpickler (it used to fall back to
    # torch.load(weights_only=False) on any exception, i.e. exactly when the allow-list had just refused the file)
    from celebrity_image_denoiser_amd import api

    with pytest.raises(pickle.UnpicklingError):
        api._read_checkpoint_file(evil)
    # a legacy (non-zip) checkpoint is the one case that goes to torch.load, with weights_only=True
    legacy = os.path.join(tmp_path, "legacy.pth")
    torch.save({"generator": {k: torch.from_numpy(v) for k, v in ref.items()}}, legacy, _use_new_zipfile_serialization=False)
    got2 = api._read_checkpoint_file(legacy)
    assert list(got2.keys()) == list(ref.keys()) and all(np.array_equal(got2[k].numpy(), ref[k]) for k in ref)


def test_trainer_checkpoint_with_numpy_scalars_loads(tmp_path):
    """the trainer's own checkpoints carry numpy.float64 scalars — `best_psnr` and every `metric_history`
    entry are np.mean(...) results (training.py:272-273,368-369,449-465) — beside real Adam / StepLR state_dicts
    (training.py:239-242,362-366).  Every loader that takes a path must open such a file, and still refuse other globals."""
    import pickle
    import zipfile

    from celebrity_image_denoiser_amd import api, ckpt

    ref = synth.make_state_dict("default")
    sd = {k: torch.from_numpy(v) for k, v in ref.items()}
    disc = torch.nn.Conv2d(3, 4, 3)
    opt = torch.optim.Adam(disc.parameters(), lr=1e-4, betas=(0.9, 0.999))
    sched = torch.optim.lr_scheduler.StepLR(opt, step_size=30, gamma=0.1)
    disc(torch.zeros(1, 3, 8, 8)).sum().backward()
    opt.step()
    sched.step()
    state = {"generator": sd, "discriminator": disc.state_dict(), "g_optimizer": opt.state_dict(), "d_optimizer": opt.state_dict(),
             "scheduler_g": sched.state_dict(), "scheduler_d": sched.state_dict(), "epoch": 499,
             "best_psnr": np.mean([30.5, 31.25]),
             "metric_history": {"psnr": [np.mean([1.0, 2.0]), np.float32(3.0)], "ssim": [np.float64(0.9)], "g_loss": [0.5], "count": [np.int64(3)]}}
    assert type(state["best_psnr"]).__module__ == "numpy"
    path = os.path.join(tmp_path, "denoise_epoch_499.pth")
    torch.save(state, path)
    whole = ckpt.read_checkpoint(path)
    assert whole["best_psnr"] == 30.875 and type(whole["best_psnr"]) is float
    assert whole["metric_history"] == {"psnr": [1.5, 3.0], "ssim": [0.9], "g_loss": [0.5], "count": [3]}
    assert whole["g_optimizer"]["param_groups"][0]["betas"] == (0.9, 0.999) and whole["scheduler_g"]["step_size"] == 30
    got = api._read_checkpoint_file(path)
    assert list(got.keys()) == list(ref.keys()) and all(np.array_equal(got[k].numpy(), ref[k]) for k in ref)
    m = cid.DenoiseGenerator()
    api.load_state_safely(m, path)               # the reference loader's name and semantics (app.py:257-274)
    assert not m.training and all(np.array_equal(m.state_dict()[k].numpy(), ref[k]) for k in ref)
    legacy = os.path.join(tmp_path, "legacy_np.pth")
    torch.save(state, legacy, _use_new_zipfile_serialization=False)
    got2 = api._read_checkpoint_file(legacy)
    assert all(np.array_equal(got2[k].numpy(), ref[k]) for k in ref)
    # the handlers are closed: an object dtype, a structured dtype or a non-latin1 encode is refused
    for bad in (ckpt._ScalarDtype, ckpt._latin1_bytes):
        with pytest.raises(pickle.UnpicklingError):
            bad("O8") if bad is ckpt._ScalarDtype else bad("x", "utf-16")
    with pytest.raises(pickle.UnpicklingError):
        ckpt._numpy_scalar(ckpt._ScalarDtype("f8"), b"\x00" * 4)
    evil = os.path.join(tmp_path, "evil_np.pth")
    with zipfile.ZipFile(evil, "w") as zf:
        zf.writestr("archive/data.pkl", pickle.dumps({"generator": {}, "best_psnr": np.array([1.0])}, protocol=2))
    with pytest.raises(pickle.UnpicklingError):   # numpy arrays (numpy._core.multiarray._reconstruct) are not scalars: still refused
        ckpt.read_checkpoint(evil)


def test_checkpoint_reader_bounds_checks_tensor_views(tmp_path):
    """size/stride/offset come from the file; a view that reaches past its storage must be refused, not read
    (as_strided without a bounds check)."""
    import pickle

    from celebrity_image_denoiser_amd import ckpt

    st = np.arange(12, dtype=np.float32)
    ok = ckpt._rebuild_tensor_v2(st, 2, (2, 3), (3, 1))
    assert np.array_equal(ok, st[2:8].reshape(2, 3))
    assert ckpt._rebuild_tensor_v2(st, 0, (0, 5), (5, 1)).shape == (0, 5)
    for off, size, stride in ((0, (4, 4), (4, 1)), (8, (2, 3), (3, 1)), (0, (2,), (12,)), (-1, (2,), (1,)), (0, (2,), (-1,)), (12, (), ())):
        with pytest.raises(pickle.UnpicklingError):
            ckpt._rebuild_tensor_v2(st, off, size, stride)


def test_host_pipeline_refuses_cpu_model():
    """The overlapped host pipeline is a GPU product path: a model that is not on a GPU is refused loudly."""
    import celebrity_image_denoiser_amd as cid

    with pytest.raises(RuntimeError, match="GPU"):
        cid.HostPipeline(cid.DenoiseGenerator())
    with pytest.raises(ValueError):
        cid.HostPipeline(cid.DenoiseGenerator(), depth=1)


def test_winograd_f4x2_transform_constants():
    """The constants k_wino42_conv hard-codes (csrc/wino42_kernels.h: B4^T rows, A4^T in w42_out4, G4 in pack_winograd42_u;
    interpolation points 0, +-3/4, +-3/2, inf) and the F(2,3) matrices of the vertical direction compute a 3x3 correlation
    exactly: Y(2x4) = A2^T [ (G2 g G4^T) . (B2^T d B4) ] A4 against the direct sum, in float64, for random g and d.  Every
    entry of B4^T and A4^T is a dyadic rational, i.e. exact in fp32 (the kernel's transforms add no coefficient rounding)."""
    B4t = np.array([[81 / 64, 0, -45 / 16, 0, 1, 0], [0, -27 / 16, -9 / 4, 3 / 4, 1, 0], [0, 27 / 16, -9 / 4, -3 / 4, 1, 0],
                    [0, -27 / 32, -9 / 16, 3 / 2, 1, 0], [0, 27 / 32, -9 / 16, -3 / 2, 1, 0], [0, 81 / 64, 0, -45 / 16, 0, 1]])
    A4t = np.array([[1, 1, 1, 1, 1, 0], [0, 3 / 4, -3 / 4, 3 / 2, -3 / 2, 0], [0, 9 / 16, 9 / 16, 9 / 4, 9 / 4, 0],
                    [0, 27 / 64, -27 / 64, 27 / 8, -27 / 8, 1]])
    G4 = np.array([[64 / 81, 0, 0], [-128 / 243, -32 / 81, -8 / 27], [-128 / 243, 32 / 81, -8 / 27],
                   [32 / 243, 16 / 81, 8 / 27], [32 / 243, -16 / 81, 8 / 27], [0, 0, 1]])
    B2t = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=np.float64)
    A2t = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=np.float64)
    G2 = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]])
    for m in (B4t, A4t):
        assert np.array_equal(m.astype(np.float32).astype(np.float64), m)      # exact in fp32
        assert np.array_equal(m * 64, np.round(m * 64))                        # dyadic, denominators <= 64
    rng = np.random.default_rng(3)
    for _ in range(10):
        g, d = rng.standard_normal((3, 3)), rng.standard_normal((4, 6))        # patch: 4 rows x 6 columns -> 2 x 4 outputs
        U = G2 @ g @ G4.T
        V = B2t @ d @ B4t.T
        Y = A2t @ (U * V) @ A4t.T
        ref = np.array([[(d[y:y + 3, x:x + 3] * g).sum() for x in range(4)] for y in range(2)])
        assert np.abs(Y - ref).max() < 1e-12


def test_tracked_profiles_describe_the_default_kernels():
    """VERDICT r2 #1 / ADVICE r2: profiles/pmc_traffic.json (read by bench.py for roofline.traffic) and the latest tracked
    rocprofv3 stats table must be profiles of the kernels the DEFAULT forward launches — the names cid_launch_kernel()
    reports — and profiles/summarize.py's own check list must agree with the library."""
    import ctypes
    import glob
    import importlib.util
    import json

    from celebrity_image_denoiser_amd import _lib

    L = _lib.lib()
    h = ctypes.c_void_p()
    assert L.cid_create(ctypes.byref(h)) == 0
    names = [L.cid_launch_kernel(h, i).decode() for i in range(_lib.CID_NUM_LAUNCHES)]
    L.cid_destroy(h)
    spec = importlib.util.spec_from_file_location("summarize", os.path.join(ROOT, "profiles", "summarize.py"))
    summ = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(summ)
    assert len(summ.EXPECT) == len(names)
    for want, name in zip(summ.EXPECT, names):
        assert want.startswith(name) or name.startswith(want), (want, name)
    traffic = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    for name in names:
        hit = [k for k in traffic if k.startswith(name)]
        assert hit and all(traffic[k] > 0 for k in hit), f"profiles/pmc_traffic.json has no entry for the default kernel {name!r}"
    latest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_final_kernel_stats.csv")))[-1]
    table = open(latest).read()
    for name in names:
        assert name in table, f"{os.path.basename(latest)} does not list the default kernel {name!r}"


def test_no_unpadded_store_data_hazard_in_the_kernels(tmp_path):
    """Round 3 found that hipcc (ROCm 7.2) lets a VALU instruction overwrite the first data register of a
    `buffer_store_dwordx4 ... s<soffset> offen` one instruction after it, and that gfx950 then stores the new value
    (profiles/r03_store_hazard.txt).  The product's stores of that form hold their data registers across an s_nop 3 (store16 in
    wino42_kernels.h); csrc/tools/store_hazard_check.py scans the generated ISA of EVERY kernel for the pattern.  Cross-compiles the
    device code (no GPU needed, about a minute)."""
    import shutil
    import subprocess
    import sys

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available: the ISA cannot be generated here")
    csrc = os.path.join(ROOT, "celebrity_image_denoiser_amd", "csrc")
    asm = os.path.join(tmp_path, "cid_kernels.s")
    subprocess.check_call([hipcc, "-O3", "-std=c++17", "-fno-slp-vectorize", "--offload-arch=gfx950", "--offload-device-only", "-S",
                           "-o", asm, os.path.join(csrc, "cid_api.hip")], cwd=csrc, stderr=subprocess.DEVNULL)
    out = subprocess.run([sys.executable, os.path.join(csrc, "tools", "store_hazard_check.py"), asm], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    n = int(out.stdout.split(":")[1].split()[0])
    assert n >= 100, out.stdout          # the scan did see the Winograd epilogue's stores


def _load_bench():
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_self_launch_starts_one_child_per_gpu_and_relays_rank0(tmp_path):
    """`python bench.py --gpus N` as a plain command (the form the driver records) must run by itself.  The
    launcher (bench.launch_ranks) starts N children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / one shared
    free MASTER_PORT, passes the command line through, relays rank 0's single stdout line and returns 0.  The child command is
    a stand-in here (no GPU); importing bench.py must not import torch or touch the library (the launcher stays GPU-free)."""
    import io
    import json
    import subprocess
    import sys

    probe = subprocess.run([sys.executable, "-c",
                            "import sys, importlib.util as u; s = u.spec_from_file_location('b', sys.argv[1]); m = u.module_from_spec(s); "
                            "s.loader.exec_module(m); print('torch' in sys.modules, 'celebrity_image_denoiser_amd' in sys.modules)",
                            os.path.join(ROOT, "bench.py")], capture_output=True, text=True, check=True)
    assert probe.stdout.split() == ["False", "False"]
    bench = _load_bench()
    child = ("import json, os, sys\n"
             "e = {k: os.environ.get(k) for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'HSA_ENABLE_IPC_MODE_LEGACY')}\n"
             "e['argv'] = sys.argv[1:]\n"
             f"json.dump(e, open(os.path.join({str(tmp_path)!r}, 'rank' + e['RANK'] + '.json'), 'w'))\n"
             "print('noise from rank ' + e['RANK'], file=sys.stderr)\n"
             "print(json.dumps({'metric': 'stub', 'n_gpus': int(e['WORLD_SIZE']), 'from_rank': int(e['RANK'])}))\n")
    out = io.StringIO()
    argv = ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    rc = bench.launch_ranks(4, argv, child_cmd=[sys.executable, "-c", child], out=out)
    assert rc == 0
    lines = [l for l in out.getvalue().splitlines() if l.strip()]
    assert len(lines) == 1 and json.loads(lines[0]) == {"metric": "stub", "n_gpus": 4, "from_rank": 0}   # only rank 0's stdout is relayed
    envs = [json.load(open(os.path.join(tmp_path, f"rank{r}.json"))) for r in range(4)]
    assert [e["RANK"] for e in envs] == ["0", "1", "2", "3"] and [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2", "3"]
    assert all(e["WORLD_SIZE"] == "4" and e["LOCAL_WORLD_SIZE"] == "4" and e["MASTER_ADDR"] == "127.0.0.1" for e in envs)
    assert len({e["MASTER_PORT"] for e in envs}) == 1 and int(envs[0]["MASTER_PORT"]) > 0
    assert all(e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and e["argv"] == argv for e in envs)


def test_bench_self_launch_propagates_a_failing_rank(tmp_path):
    """A rank that exits non-zero makes the launcher return that code promptly and stop the other ranks (they would wait for
    it inside a collective); nothing of a failed run is mistaken for a result line."""
    import io
    import sys
    import time

    bench = _load_bench()
    child = ("import os, sys, time\n"
             "r = int(os.environ['RANK'])\n"
             f"open(os.path.join({str(tmp_path)!r}, 'started%d' % r), 'w').close()\n"
             "if r == 2:\n"
             "    time.sleep(0.5); sys.exit(7)\n"
             "time.sleep(120)\n"
             "print('{\"never\": 1}')\n")
    out = io.StringIO()
    t0 = time.time()
    rc = bench.launch_ranks(3, [], child_cmd=[sys.executable, "-c", child], out=out)
    assert rc == 7 and time.time() - t0 < 60 and out.getvalue().strip() == ""
    assert all(os.path.exists(os.path.join(tmp_path, f"started{r}")) for r in range(3))


def test_bench_plain_command_with_gpus_n_uses_the_launcher(monkeypatch):
    """main(): --gpus N > 1 without WORLD_SIZE goes to launch_ranks with the untouched command line and exits with its code;
    with WORLD_SIZE set (a rank started by torch.distributed.run or by the launcher) it does not."""
    import sys

    bench = _load_bench()
    seen = {}

    def fake(n, argv, **kw):
        seen.update(n=n, argv=list(argv))
        return 5

    monkeypatch.setattr(bench, "launch_ranks", fake)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 5 and seen == {"n": 8, "argv": ["--gpus", "8", "--steps", "20", "--warmup", "5"]}
    seen.clear()
    monkeypatch.setenv("WORLD_SIZE", "8")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setattr(bench, "_import_runtime", lambda: (_ for _ in ()).throw(KeyboardInterrupt("reached the rank path")))
    with pytest.raises(KeyboardInterrupt):
        bench.main()
    assert seen == {}
