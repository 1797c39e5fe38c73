"""ctypes binding of include/cid.h -> libcid.so.  No fallback: a missing library is an error."""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CID_LIB_PATH: development aid for same-box A/B runs of two builds of the library (bench.py in alternation)
LIB_PATH = os.environ.get("CID_LIB_PATH") or os.path.join(_HERE, "libcid.so")

CID_OK = 0
CID_NUM_PARAMS = 24
CID_NUM_LAUNCHES = 12
CID_ALGO_DIRECT, CID_ALGO_WINOGRAD64, CID_ALGO_WINOGRAD42, CID_ALGO_SPLIT16 = 0, 2, 3, 4
CID_FMT_F32_NCHW, CID_FMT_U8_NHWC = 0, 1
CID_DTYPE_F32, CID_DTYPE_F16 = 0, 1
CID_TAIL_FUSED, CID_TAIL_BANDS, CID_TAIL_TILES = 0, 1, 2

# every symbol include/cid.h declares: (restype, argtypes)
_c = ctypes
SYMBOLS = {
    "cid_version": (_c.c_char_p, []),
    "cid_create": (_c.c_int, [_c.POINTER(_c.c_void_p)]),
    "cid_destroy": (None, [_c.c_void_p]),
    "cid_last_error": (_c.c_char_p, [_c.c_void_p]),
    "cid_set_weight": (_c.c_int, [_c.c_void_p, _c.c_char_p, _c.c_void_p, _c.POINTER(_c.c_int64), _c.c_int]),
    "cid_get_weight": (_c.c_int, [_c.c_void_p, _c.c_char_p, _c.c_void_p, _c.c_size_t]),
    "cid_missing_weights": (_c.c_int, [_c.c_void_p, _c.POINTER(_c.c_int)]),
    "cid_param_key": (_c.c_char_p, [_c.c_int]),
    "cid_packed_weights_bytes": (_c.c_size_t, []),
    "cid_upload_weights": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p]),
    "cid_export_packed": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_size_t]),
    "cid_import_packed": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_size_t]),
    "cid_attach_weights": (_c.c_int, [_c.c_void_p, _c.c_void_p]),
    "cid_out_shape": (_c.c_int, [_c.c_int, _c.c_int, _c.POINTER(_c.c_int), _c.POINTER(_c.c_int)]),
    "cid_workspace_bytes": (_c.c_int, [_c.c_int, _c.c_int, _c.c_int, _c.POINTER(_c.c_size_t)]),
    "cid_forward": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int,
                               _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "cid_forward_ex": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_int,
                                  _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "cid_forward_padded": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_int,
                                      _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p, _c.c_size_t, _c.c_void_p]),
    "cid_view_u8": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_void_p]),
    "cid_forward_timed": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_int,
                                     _c.c_void_p, _c.c_size_t, _c.c_void_p, _c.POINTER(_c.c_float)]),
    "cid_timing_begin": (_c.c_int, [_c.c_void_p, _c.c_int]),
    "cid_timing_end": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.POINTER(_c.c_float), _c.POINTER(_c.c_int)]),
    "cid_launch_name": (_c.c_char_p, [_c.c_int]),
    "cid_launch_kernel": (_c.c_char_p, [_c.c_void_p, _c.c_int]),
    "cid_set_conv_algo": (_c.c_int, [_c.c_void_p, _c.c_int]),
    "cid_get_conv_algo": (_c.c_int, [_c.c_void_p, _c.POINTER(_c.c_int)]),
    "cid_set_tail_algo": (_c.c_int, [_c.c_void_p, _c.c_int]),
    "cid_get_tail_algo": (_c.c_int, [_c.c_void_p, _c.POINTER(_c.c_int)]),
    "cid_debug_poison_lds": (_c.c_int, [_c.c_void_p]),
    "cid_debug_winograd_workgroups_per_cu": (_c.c_int, [_c.c_int]),
    "cid_debug_half_workgroups_per_cu": (_c.c_int, [_c.c_int]),
    "cid_debug_winograd_column_block_per_xcd": (_c.c_int, [_c.c_int]),
    "cid_set_compute_dtype": (_c.c_int, [_c.c_void_p, _c.c_int]),
    "cid_get_compute_dtype": (_c.c_int, [_c.c_void_p, _c.POINTER(_c.c_int)]),
    "cid_launch_work_ex": (_c.c_int, [_c.c_void_p, _c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.POINTER(_c.c_double), _c.POINTER(_c.c_double)]),
    "cid_stage_view": (_c.c_int, [_c.c_char_p, _c.c_int, _c.c_int, _c.c_int, _c.POINTER(_c.c_size_t), _c.POINTER(_c.c_int),
                                  _c.POINTER(_c.c_int), _c.POINTER(_c.c_int), _c.POINTER(_c.c_int), _c.POINTER(_c.c_int)]),
    "cid_comm_available": (_c.c_int, []),
    "cid_comm_unique_id": (_c.c_int, [_c.c_void_p]),
    "cid_comm_init_rank": (_c.c_int, [_c.POINTER(_c.c_void_p), _c.c_int, _c.c_void_p, _c.c_int]),
    "cid_comm_destroy": (_c.c_int, [_c.c_void_p]),
    "cid_comm_count": (_c.c_int, [_c.c_void_p, _c.POINTER(_c.c_int)]),
    "cid_broadcast_weights": (_c.c_int, [_c.c_void_p, _c.c_void_p, _c.c_int, _c.c_int, _c.c_void_p]),
    "cid_launch_work": (_c.c_int, [_c.c_int, _c.c_int, _c.c_int, _c.c_int, _c.POINTER(_c.c_double), _c.POINTER(_c.c_double)]),
}

_lib = None


class CidError(RuntimeError):
    """A cid_* call returned a non-zero status."""

    def __init__(self, code: int, message: str):
        super().__init__(message)
        self.code = code


def lib() -> ctypes.CDLL:
    """Load libcid.so (built by `python -c 'import __graft_entry__ as g; g.build()'` or csrc/Makefile)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension is not built. This package has no CPU or PyTorch "
                "fallback; build it with `make -C celebrity_image_denoiser_amd/csrc` (needs hipcc)."
            )
        # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64.so (same SONAME as
        # /opt/rocm's).  Load torch's copy first so libcid.so binds to the runtime that owns the
        # tensors and streams it is handed; two runtimes in one process do not see each other's devices.
        import torch

        hip_rt = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        if os.path.exists(hip_rt):
            ctypes.CDLL(hip_rt, mode=ctypes.RTLD_GLOBAL)
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError here = header and library out of sync
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(handle, code: int):
    if code != CID_OK:
        msg = lib().cid_last_error(handle)
        raise CidError(code, (msg.decode() if msg else "") or f"cid error {code}")
