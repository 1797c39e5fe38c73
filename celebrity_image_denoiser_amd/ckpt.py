"""Read a PyTorch `.pth` checkpoint without importing torch.

The reference's trainer writes `torch.save({"generator": state_dict, "discriminator": ..., "g_optimizer": ...,
"epoch": ..., ...})` (backend/trainingcode/denoise_gan_code/training.py:359-376) and the server reads it with
`torch.load` (backend/app.py:257-274).  A `.pth` written by torch >= 1.6 is a zip archive: `<name>/data.pkl`
(a pickle whose tensors are persistent ids pointing at storages) and `<name>/data/<key>` (raw little-endian
storage bytes).  This reader unpickles with a closed allow-list of globals — nothing from the file can run code —
and returns tensors as numpy arrays.  Non-tensor entries (epoch, metric history, optimizer hyper-parameters) come
back as plain Python objects.
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
