"""Read a PyTorch `.pth` checkpoint without importing torch.

The reference's trainer writes `torch.save({"generator": state_dict, "discriminator": ..., "g_optimizer": ...,
"epoch": ..., ...})` (backend/trainingcode/denoise_gan_code/training.py:359-376) and the server reads it with
`torch.load` (backend/app.py:257-274).  A `.pth` written by torch >= 1.6 is a zip archive: `<name>/data.pkl`
(a pickle whose tensors are persistent ids pointing at storages) and `<name>/data/<key>` (raw little-endian
storage bytes).  This reader unpickles with a closed allow-list of globals — nothing from the file can run code —
and returns tensors as numpy arrays.  Non-tensor entries (epoch, metric history, optimizer hyper-parameters) come
back as plain Python objects.

The trainer's `best_psnr` and `metric_history` values are `np.mean(...)` results (training.py:272-273, 449-465), i.e.
numpy.float64 scalars, which pickle as `numpy._core.multiarray.scalar(numpy.dtype("f8"), <8 bytes>)` (numpy >= 2;
`numpy.core.multiarray.scalar` before).  Those two globals are answered by closed handlers here — a fixed table of
numeric dtype codes, the bytes decoded into a Python int/float/bool — so real `denoise_epoch_*.pth` files load and
still nothing from the file is called.
"""
from __future__ import annotations

import pickle
import zipfile
from collections import OrderedDict

import numpy as np

_DTYPES = {
    "FloatStorage": np.float32, "DoubleStorage": np.float64, "HalfStorage": np.float16,
    "LongStorage": np.int64, "IntStorage": np.int32, "ShortStorage": np.int16,
    "CharStorage": np.int8, "ByteStorage": np.uint8, "BoolStorage": np.bool_,
}


class _StorageType:
    def __init__(self, name):
        self.name = name
        self.dtype = _DTYPES[name]


def _rebuild_tensor_v2(storage, storage_offset, size, stride, requires_grad=False, backward_hooks=None, metadata=None):
    size, stride = tuple(int(v) for v in size), tuple(int(v) for v in stride)
    storage_offset = int(storage_offset)
    # size/stride/offset come from the file: every element they address must lie inside the storage
    if len(size) != len(stride) or storage_offset < 0 or any(v < 0 for v in size) or any(v < 0 for v in stride):
        raise pickle.UnpicklingError("tensor with negative or inconsistent size/stride/offset in checkpoint")
    if all(v > 0 for v in size):
        last = storage_offset + sum((n - 1) * st for n, st in zip(size, stride))
        if last >= storage.size:
            raise pickle.UnpicklingError("tensor view reaches past the end of its storage")
    if len(size) == 0:
        return np.array(storage[storage_offset], dtype=storage.dtype)
    if any(v == 0 for v in size):
        return np.empty(size, dtype=storage.dtype)
    itemsize = storage.dtype.itemsize
    view = np.lib.stride_tricks.as_strided(storage[storage_offset:], shape=size, strides=tuple(s * itemsize for s in stride))
    return np.array(view)   # own, contiguous copy


def _rebuild_parameter(data, requires_grad=False, backward_hooks=None):
    return data


# numpy scalars inside a checkpoint (see the module docstring): dtype code -> numpy type, nothing else is accepted
_SCALAR_CODES = {
    "f8": np.float64, "f4": np.float32, "f2": np.float16, "i8": np.int64, "i4": np.int32, "i2": np.int16, "i1": np.int8,
    "u8": np.uint64, "u4": np.uint32, "u2": np.uint16, "u1": np.uint8, "b1": np.bool_,
}


class _ScalarDtype:
    """Stand-in for `numpy.dtype(code, align, copy)` + its pickled state `(version, byteorder, ...)`."""

    def __init__(self, code, *_ignored):
        if not isinstance(code, str) or code not in _SCALAR_CODES:
            raise pickle.UnpicklingError(f"checkpoint holds a numpy scalar of dtype {code!r}, which this reader does not allow")
        self.dtype = np.dtype(_SCALAR_CODES[code])

    def __setstate__(self, state):
        order = state[1] if isinstance(state, tuple) and len(state) > 1 else "="
        if order not in ("<", ">", "=", "|"):
            raise pickle.UnpicklingError("numpy dtype with an unknown byte order in checkpoint")
        if order in ("<", ">"):
            self.dtype = self.dtype.newbyteorder(order)


def _latin1_bytes(text, encoding="latin1"):
    """`_codecs.encode(str, "latin1")`: how pickle protocol 2 (torch.save's default) spells a bytes object."""
    if not isinstance(text, str) or encoding != "latin1" or len(text) > 16:
        raise pickle.UnpicklingError("unexpected _codecs.encode call in checkpoint")
    return text.encode("latin1")


def _numpy_scalar(dtype, raw):
    if not isinstance(dtype, _ScalarDtype) or not isinstance(raw, (bytes, bytearray)) or len(raw) != dtype.dtype.itemsize:
        raise pickle.UnpicklingError("malformed numpy scalar in checkpoint")
    return np.frombuffer(bytes(raw), dtype=dtype.dtype)[0].item()   # a plain Python number


class _Unpickler(pickle.Unpickler):
    def __init__(self, f, zf, prefix):
        super().__init__(f)
        self._zf, self._prefix, self._cache = zf, prefix, {}

    def find_class(self, module, name):
        if module == "collections" and name == "OrderedDict":
            return OrderedDict
        if module == "torch._utils" and name == "_rebuild_tensor_v2":
            return _rebuild_tensor_v2
        if module == "torch._utils" and name == "_rebuild_parameter":
            return _rebuild_parameter
        if module == "torch" and name in _DTYPES:
            return _StorageType(name)
        if module in ("numpy", "numpy.core", "numpy._core") and name == "dtype":
            return _ScalarDtype
        if module in ("numpy.core.multiarray", "numpy._core.multiarray") and name == "scalar":
            return _numpy_scalar
        if module == "_codecs" and name == "encode":
            return _latin1_bytes
        if module == "builtins" and name in ("dict", "list", "tuple", "set", "int", "float", "str", "bool"):
            return __import__("builtins").__dict__[name]
        raise pickle.UnpicklingError(f"checkpoint references {module}.{name}, which this reader does not allow")

    def persistent_load(self, pid):
        # ('storage', storage_type, key, location, numel)
        if not (isinstance(pid, tuple) and pid and pid[0] == "storage"):
            raise pickle.UnpicklingError("unsupported persistent id in checkpoint")
        _, stype, key, _location, numel = pid
        if key not in self._cache:
            raw = self._zf.read(f"{self._prefix}/data/{key}")
            arr = np.frombuffer(raw, dtype=stype.dtype)
            if arr.size < numel:
                raise pickle.UnpicklingError("storage shorter than recorded")
            self._cache[key] = arr
        return self._cache[key]


def read_checkpoint(path: str):
    """-> the object that was saved (dicts / OrderedDicts / lists / numbers), tensors as numpy arrays."""
    with zipfile.ZipFile(path) as zf:
        pkl = [n for n in zf.namelist() if n.endswith("/data.pkl")]
        if len(pkl) != 1:
            raise ValueError(f"{path}: not a torch>=1.6 zip checkpoint (legacy formats need torch.load)")
        prefix = pkl[0][: -len("/data.pkl")]
        with zf.open(pkl[0]) as f:
            return _Unpickler(f, zf, prefix).load()


def read_state_dict(path: str, key_candidates=("generator", "state_dict", "G")) -> "OrderedDict[str, np.ndarray]":
    """Checkpoint file -> flat {key: float32 array} with the reference loader's unwrapping rules
    (backend/app.py:259-271: sub-dict under "generator"/"state_dict"/"G", strip a leading "module.")."""
    ckpt = read_checkpoint(path)
    state = ckpt
    if isinstance(ckpt, dict):
        for k in key_candidates:
            if k in ckpt and isinstance(ckpt[k], dict):
                state = ckpt[k]
                break
    out = OrderedDict()
    for k, v in state.items():
        if isinstance(k, str) and k.startswith("module."):
            k = k.replace("module.", "")
        out[k] = v
    return out
