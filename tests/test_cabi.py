"""No-GPU checks of the C ABI: libcid.so loads, exports every symbol include/cid.h declares, and its
host-only entry points (weight staging, shape/workspace planning, work tables) behave.  No compute call."""
import ctypes
import os
import re

import numpy as np
import pytest

from celebrity_image_denoiser_amd import _lib, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "cid.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(cid_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_header_symbol():
    names = _declared_symbols()
    assert len(names) >= 20
    L = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/cid.h but not exported by libcid.so"
    assert set(names) == set(_lib.SYMBOLS), "python binding table and header disagree"


def test_version_and_tables():
    L = _lib.lib()
    assert b"gfx950" in L.cid_version()
    keys = [L.cid_param_key(i).decode() for i in range(_lib.CID_NUM_PARAMS)]
    assert keys == list(synth.param_shapes().keys())          # reference state_dict order, app.py:42-78
    assert L.cid_param_key(24) is None
    assert [L.cid_launch_name(i).decode() for i in range(12)] == [name for name, *_ in synth.LAYERS]


def test_out_shape_and_workspace():
    L = _lib.lib()
    ho, wo = ctypes.c_int(), ctypes.c_int()
    for (h, w), exp in {(128, 128): (128, 128), (127, 130): (124, 128), (13, 18): (12, 16), (4, 4): (4, 4), (7, 9): (4, 8)}.items():
        assert L.cid_out_shape(h, w, ctypes.byref(ho), ctypes.byref(wo)) == 0
        assert (ho.value, wo.value) == exp
    assert L.cid_out_shape(3, 8, ctypes.byref(ho), ctypes.byref(wo)) == 2      # CID_ERR_SHAPE
    n = ctypes.c_size_t()
    assert L.cid_workspace_bytes(256, 128, 128, ctypes.byref(n)) == 0
    per_img = 128 * 128 * (64 + 128 + 64) + 64 * 64 * (64 + 128 + 256 + 128 + 128) + 32 * 32 * (128 + 256 + 256)
    assert n.value == 256 * per_img * 4                                          # NHWC fp32 arena, DESIGN.md
    assert L.cid_workspace_bytes(0, 128, 128, ctypes.byref(n)) == 2


def test_oversize_image_is_an_error_not_wrong_borders():
    """A single image whose widest per-image activation ([4*(H/4), 4*(W/4), 128] fp32) reaches the kernels' zero-padding
    sentinel offset 0x7ffffff0 must be refused (2048x2048 was accepted and read real data as padding).
    No GPU is touched: the shape check comes before any HIP call."""
    L = _lib.lib()
    n = ctypes.c_size_t()
    for (h, w) in ((2048, 2048), (2304, 2048), (2400, 2000), (4, 1 << 20)):
        assert L.cid_workspace_bytes(1, h, w, ctypes.byref(n)) == 2, (h, w)
    assert L.cid_workspace_bytes(1, 2047, 2048, ctypes.byref(n)) == 0           # 2047*2048*512 < 0x7ffffff0
    assert L.cid_workspace_bytes(1, 2000, 2000, ctypes.byref(n)) == 0
    # tile decode by multiply-high: N x t^2 < 2^32 with t = the most tiles per image of any launch (128x128: 4 x 32 = 128 tiles of 32x4)
    assert L.cid_workspace_bytes(262144, 128, 128, ctypes.byref(n)) == 2
    assert L.cid_workspace_bytes(262143, 128, 128, ctypes.byref(n)) == 0
    h = ctypes.c_void_p()
    assert L.cid_create(ctypes.byref(h)) == 0
    fake = ctypes.c_void_p(1 << 20)                                              # aligned, never dereferenced
    assert L.cid_attach_weights(h, fake) == 0
    rc = L.cid_forward(h, fake, fake, 1, 2048, 2048, fake, 1 << 40, None)
    assert rc == 2 and b"too large" in L.cid_last_error(h) and b"stripes" in L.cid_last_error(h)
    rc = L.cid_forward(h, fake, fake, 1, 3, 3, fake, 1 << 40, None)
    assert rc == 2 and b"too small" in L.cid_last_error(h)
    # cid_forward_padded (the server's pad -> network -> crop in one call): the crop window must exist in the network's output
    pad = lambda H, W, l, t, r, b: L.cid_forward_padded(h, fake, 1, fake, 1, 1, H, W, l, t, r, b, fake, 1 << 40, None)  # noqa: E731
    assert pad(30, 45, 0, 3, 1, 0) == 2 and b"does not fit" in L.cid_last_error(h)      # 33 x 46 -> output 32 x 44
    assert pad(37, 50, 0, 0, 0, 0) == 2                                                  # unpadded 37 x 50 -> output 36 x 48
    assert pad(30, 45, -1, 1, 2, 1) == 1 and pad(30, 45, 1, 1, 2, 5000) == 1             # bad paddings
    assert pad(1, 1, 0, 0, 1, 1) == 2 and b"too small" in L.cid_last_error(h)            # padded image 2 x 2
    assert pad(2040, 2040, 4, 4, 4, 4) == 2 and b"too large" in L.cid_last_error(h)      # the PADDED size counts for the one-call limit
    # cid_view_u8 (the reference's *0.5+0.5 / clamp / ToPILImage view as a stand-alone pass): argument checks only, no launch
    assert L.cid_view_u8(None, fake, 1, 4, 4, None) == 1 and L.cid_view_u8(fake, None, 1, 4, 4, None) == 1
    assert L.cid_view_u8(fake, fake, 0, 4, 4, None) == 1 and L.cid_view_u8(fake, fake, 1, 0, 4, None) == 1
    L.cid_destroy(h)


def test_algorithmic_work_matches_survey():
    """11,521,753,088 FLOP per 128x128 image, 46,087,012,352 per 256x256."""
    L = _lib.lib()
    f, b = ctypes.c_double(), ctypes.c_double()
    tot = 0.0
    for i in range(12):
        assert L.cid_launch_work(i, 1, 128, 128, ctypes.byref(f), ctypes.byref(b)) == 0
        tot += f.value
    assert tot == 11521753088.0
    tot = sum((L.cid_launch_work(i, 1, 256, 256, ctypes.byref(f), ctypes.byref(b)), f.value)[1] for i in range(12))
    assert tot == 46087012352.0
    L.cid_launch_work(0, 1, 128, 128, ctypes.byref(f), ctypes.byref(b))
    assert b.value == 4390912 + (3 * 64 * 9 + 64) * 4      # a1: 4,390,912 activation bytes/img + weights


def test_stage_views_and_fused_launch_accounting():
    """cid_stage_view: where the reference module's hook outputs lie in the workspace (per-stage GPU parity reads them back);
    cid_launch_work_ex: per-LAUNCH work under the handle's configuration — with the default fused last layer, launch 10
    carries upconv1[0] + upconv1[2] and writes 27 planes, launch 11 only sums; the totals over the forward do not change."""
    L = _lib.lib()
    off, c, hs, ws, ps, co = ctypes.c_size_t(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    args = (ctypes.byref(off), ctypes.byref(c), ctypes.byref(hs), ctypes.byref(ws), ctypes.byref(ps), ctypes.byref(co))
    assert L.cid_stage_view(b"down1", 2, 13, 18, *args) == 0
    assert (c.value, hs.value, ws.value, ps.value, co.value) == (64, 12, 16, 128, 64)      # skip tensor: cropped, upper half of cat1
    t0_floats = 2 * 13 * 18 * 64
    assert off.value == ((t0_floats + 63) // 64 * 64) * 4                                    # cat1 follows t0 in the arena
    assert L.cid_stage_view(b"bottleneck", 2, 13, 18, *args) == 0 and (c.value, hs.value, ws.value, ps.value, co.value) == (256, 3, 4, 256, 0)
    assert L.cid_stage_view(b"up1", 2, 13, 18, *args) == 0 and (c.value, co.value) == (64, 0)
    assert L.cid_stage_view(b"upconv1", 2, 13, 18, *args) == 3                               # fused into the last kernel: never stored
    assert L.cid_stage_view(b"down1", 1, 3, 3, *args) == 2
    h = ctypes.c_void_p()
    assert L.cid_create(ctypes.byref(h)) == 0
    f, b, fe, be = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    algo = ctypes.c_int()
    assert L.cid_get_tail_algo(h, ctypes.byref(algo)) == 0 and algo.value == _lib.CID_TAIL_FUSED
    tot = {}
    for mode in (_lib.CID_TAIL_FUSED, _lib.CID_TAIL_BANDS):
        assert L.cid_set_tail_algo(h, mode) == 0
        tot[mode] = [0.0, 0.0]
        for i in range(12):
            assert L.cid_launch_work_ex(h, i, 256, 128, 128, ctypes.byref(fe), ctypes.byref(be)) == 0
            assert L.cid_launch_work(i, 256, 128, 128, ctypes.byref(f), ctypes.byref(b)) == 0
            if mode == _lib.CID_TAIL_BANDS or i < 10:
                pooled = {1: 64 * 64 * 64, 3: 32 * 32 * 128}.get(i, 0)     # launches 1 / 3 also write pool1 / pool2 (fused MaxPool2d)
                assert (fe.value, be.value) == (f.value, b.value + 4.0 * 256 * pooled)
            tot[mode][0] += fe.value
            tot[mode][1] += be.value
            if mode == _lib.CID_TAIL_FUSED and i == 11:
                assert fe.value == 0.0 and be.value == 4.0 * (256 * 128 * 128 * 30 + 3)    # 27 z planes in, 3 channels out, bias
    assert tot[_lib.CID_TAIL_FUSED][0] == tot[_lib.CID_TAIL_BANDS][0] == 256 * 11521753088.0
    px = 256 * 128 * 128
    assert tot[_lib.CID_TAIL_BANDS][1] - tot[_lib.CID_TAIL_FUSED][1] == 4.0 * px * 2 * (64 - 27)   # the 64-channel tensor never exists
    assert L.cid_set_tail_algo(h, 7) == 1
    assert L.cid_set_conv_algo(h, 1) == 1 and b"unknown algorithm" in L.cid_last_error(h)   # round 1's first Winograd kernel is gone
    import ctypes as _ct
    algo = _ct.c_int(-1)
    assert L.cid_get_conv_algo(h, _ct.byref(algo)) == 0 and algo.value == _lib.CID_ALGO_WINOGRAD42   # the default: Winograd F(4x2,3x3)
    for a in (_lib.CID_ALGO_DIRECT, _lib.CID_ALGO_WINOGRAD64, _lib.CID_ALGO_WINOGRAD42):
        assert L.cid_set_conv_algo(h, a) == 0 and L.cid_get_conv_algo(h, _ct.byref(algo)) == 0 and algo.value == a
        assert L.cid_launch_kernel(h, 5).decode().startswith({_lib.CID_ALGO_DIRECT: "k_gemm_conv<256, 256", _lib.CID_ALGO_WINOGRAD64: "k_wino64_conv<256, 256",
                                                              _lib.CID_ALGO_WINOGRAD42: "k_wino42_conv<256, 256"}[a])
    L.cid_destroy(h)


def test_weight_staging_roundtrip_and_errors():
    L = _lib.lib()
    h = ctypes.c_void_p()
    assert L.cid_create(ctypes.byref(h)) == 0
    miss = ctypes.c_int()
    L.cid_missing_weights(h, ctypes.byref(miss))
    assert miss.value == 24
    sd = synth.make_state_dict("hot")
    for k, v in sd.items():
        shape = (ctypes.c_int64 * v.ndim)(*v.shape)
        assert L.cid_set_weight(h, k.encode(), v.ctypes.data, shape, v.ndim) == 0
    L.cid_missing_weights(h, ctypes.byref(miss))
    assert miss.value == 0
    for k, v in sd.items():                                   # unpack(pack(w)) == w for every layer layout
        out = np.empty_like(v)
        assert L.cid_get_weight(h, k.encode(), out.ctypes.data, out.size) == 0
        assert np.array_equal(out, v), k
    w = sd["up2.weight"]
    bad = (ctypes.c_int64 * 4)(128, 256, 2, 2)               # Conv2d-style dims for a ConvTranspose2d weight
    assert L.cid_set_weight(h, b"up2.weight", w.ctypes.data, bad, 4) == 2
    assert b"size mismatch for up2.weight" in L.cid_last_error(h)
    assert L.cid_set_weight(h, b"down3.0.weight", w.ctypes.data, bad, 4) == 3
    assert b"unexpected key" in L.cid_last_error(h)
    # packed blob export/import carries the same tensors to another handle
    nbytes = L.cid_packed_weights_bytes()
    blob = np.empty(nbytes, np.uint8)
    assert L.cid_export_packed(h, blob.ctypes.data, nbytes) == 0
    h2 = ctypes.c_void_p()
    L.cid_create(ctypes.byref(h2))
    assert L.cid_import_packed(h2, blob.ctypes.data, nbytes) == 0
    assert L.cid_import_packed(h2, blob.ctypes.data, nbytes - 4) == 2
    out = np.empty_like(sd["bottleneck.2.weight"])
    L.cid_get_weight(h2, b"bottleneck.2.weight", out.ctypes.data, out.size)
    assert np.array_equal(out, sd["bottleneck.2.weight"])
    # forward without device weights is refused with a message, not a crash (no GPU touched)
    dummy = np.zeros(16, np.float32)
    rc = L.cid_forward(h2, dummy.ctypes.data, dummy.ctypes.data, 1, 8, 8, dummy.ctypes.data, 64, None)
    assert rc == 4 and b"no device weights" in L.cid_last_error(h2)
    L.cid_destroy(h)
    L.cid_destroy(h2)


def test_packed_layout_of_a_gemm_layer():
    """Spot-check the documented packed order [nb][chunk][tap][g][ns][lane][e] against the reference
    tensor for down2.0 (Conv2d 64->128): lane (h, j) of step (chunk, tap, g) holds
    W[co = 64 nb + 32 ns + j][ci = 32 chunk + 8 g + 4 h + e][kh][kw]."""
    L = _lib.lib()
    h = ctypes.c_void_p()
    L.cid_create(ctypes.byref(h))
    w = synth.make_state_dict("default")["down2.0.weight"]
    shape = (ctypes.c_int64 * 4)(*w.shape)
    assert L.cid_set_weight(h, b"down2.0.weight", w.ctypes.data, shape, 4) == 0
    blob = np.empty(L.cid_packed_weights_bytes(), np.uint8)
    L.cid_export_packed(h, blob.ctypes.data, blob.nbytes)
    f = blob.view(np.float32)
    # offset of layer 2's weights: after head (w 1792 -> 1792, b 64) and down1.2 (w 36864, b 64), 64-float aligned
    off = 1792 + 64 + 36864 + 64
    nb, ck, tap, g, ns, hh, j, e = 1, 1, 5, 2, 1, 1, 17, 3
    idx = ((((nb * 2 + ck) * 9 + tap) * 4 + g) * 2 + ns) * 256 + (hh * 32 + j) * 4 + e
    assert f[off + idx] == w[64 * nb + 32 * ns + j, 32 * ck + 8 * g + 4 * hh + e, tap // 3, tap % 3]
    L.cid_destroy(h)
