"""MI355X-native forward of the reference's denoising U-Net (`DenoiseGenerator`).

Public surface (mirrors what the reference's callers use, reference backend/app.py:319-336,422-435):

    DenoiseGenerator()            nn.Module-protocol object: .to(), .load_state_dict(), .eval(), __call__
    load(path_or_state_dict)      -> DenoiseGenerator on the current GPU, weights loaded like load_state_safely
    denoise(model, image_batch)   -> image_batch
    denoise_u8(model, uint8 NHWC) -> uint8 NHWC (pre/post-processing fused into the first/last kernel)
    HostPipeline(model).run(host_batches)   upload / forward / download overlapped on three HIP streams
    GraphedForward(model, example)(x)       the forward at a fixed shape as one HIP-graph launch (N=1 serving latency)
    enhance_images(ckpt, in_dir, out_dir)   the reference's directory eval harnesses (denoisegan_eval.py / denoise_eavl_iter.py)

Everything numeric runs in hand-written HIP kernels behind the C ABI in include/cid.h
(csrc/ -> libcid.so).  There is no CPU fallback: if the library is missing the calls raise.
"""
__version__ = "0.1.0"

_LAZY = {
    "DenoiseGenerator": ("generator", "DenoiseGenerator"),
    "load": ("api", "load"),
    "denoise": ("api", "denoise"),
    "denoise_u8": ("api", "denoise_u8"),
    "serve_u8": ("api", "serve_u8"),
    "get_padding": ("api", "get_padding"),
    "load_state_safely": ("api", "load_state_safely"),
    "psnr": ("metrics", "psnr"),
    "HostPipeline": ("pipeline", "HostPipeline"),
    "denoise_host_batches": ("pipeline", "denoise_host_batches"),
    "GraphedForward": ("pipeline", "GraphedForward"),
    "enhance_images": ("harness", "enhance_images"),
}


def __getattr__(name):
    if name in _LAZY:
        import importlib

        mod, attr = _LAZY[name]
        return getattr(importlib.import_module(__name__ + "." + mod), attr)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
