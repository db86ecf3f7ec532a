"""dtype tokens and container rules -- mirrors taichi_image/types.py:12-78 of the reference.

The reference passes Taichi dtype objects (``ti.u8`` ...).  Taichi is not a dependency here, so
this module defines equivalent tokens (``u8, u16, f16, f32`` and their long aliases) and accepts
numpy dtypes, torch dtypes, strings and Taichi dtypes (matched by name) wherever the reference
takes a ``dtype`` argument.  "Output lives where the input lives" (types.py:70-78): numpy in ->
numpy out, torch in -> torch out on the same device; the computation itself always runs on the
MI355X, host containers are staged through HBM.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _native


class DType:
    __slots__ = ("name", "code", "np", "torch", "scale")

    def __init__(self, name, code, np_dtype, torch_dtype, scale):
        self.name, self.code, self.np, self.torch, self.scale = name, code, np_dtype, torch_dtype, scale

    def __repr__(self):
        return self.name


u8 = uint8 = DType("u8", _native.MI_U8, np.uint8, torch.uint8, 255)
u16 = uint16 = DType("u16", _native.MI_U16, np.uint16, torch.uint16, 65535)
f16 = float16 = DType("f16", _native.MI_F16, np.float16, torch.float16, 1.0)
f32 = float32 = DType("f32", _native.MI_F32, np.float32, torch.float32, 1.0)

_ALL = (u8, u16, f16, f32)

# types.py:12-18
scale_factor = {d: d.scale for d in _ALL}
# types.py:21-33
ti_to_np = {d: d.np for d in _ALL}
ti_to_torch = {u8: torch.uint8, f16: torch.float16, f32: torch.float32}

_by_name = {}
for _d in _ALL:
    for _n in (_d.name, np.dtype(_d.np).name, str(_d.torch), str(_d.torch).replace("torch.", "")):
        _by_name[_n] = _d
# Taichi spellings (str(ti.u8) == 'u8', ti.uint8 is ti.u8, ...)
_by_name.update({"uint8": u8, "uint16": u16, "float16": f16, "float32": f32, "half": f16, "float": f32})


def as_dtype(d) -> DType:
    """Normalise any accepted dtype spelling to a token; KeyError when unsupported
    (the reference raises KeyError from its lookup tables, types.py:53-55)."""
    if isinstance(d, DType):
        return d
    if isinstance(d, torch.dtype):
        key = str(d)
    elif isinstance(d, (np.dtype, type)):
        try:
            key = np.dtype(d).name
        except TypeError:
            key = str(d)
    else:
        key = str(d)
    if key not in _by_name:
        raise KeyError(f"unsupported dtype {d!r}")
    return _by_name[key]


def ti_type(in_arr) -> DType:
    """types.py:51-57."""
    if isinstance(in_arr, np.ndarray):
        return as_dtype(in_arr.dtype)
    if isinstance(in_arr, torch.Tensor):
        return as_dtype(in_arr.dtype)
    raise ValueError(f"Unsupported input type {type(in_arr)}")


def empty_like(in_arr, shape=None, dtype=None):
    """types.py:70-78: same container kind and device as the input."""
    shape = in_arr.shape if shape is None else shape
    dtype = ti_type(in_arr) if dtype is None else as_dtype(dtype)
    if isinstance(in_arr, np.ndarray):
        return np.empty(tuple(shape), dtype.np)
    if isinstance(in_arr, torch.Tensor):
        return torch.empty(tuple(shape), dtype=dtype.torch, device=in_arr.device)
    raise ValueError(f"Unsupported input type {type(in_arr)}")


def zeros_like(in_arr, shape=None, dtype=None):
    """types.py:81-91."""
    out = empty_like(in_arr, shape, dtype)
    if isinstance(out, np.ndarray):
        out[...] = 0
    else:
        out.zero_()
    return out


# ---- staging between the caller's container and HBM ----------------------------------------------

def default_device() -> torch.device:
    _native.require_gpu()
    return torch.device("cuda", torch.cuda.current_device())


def to_device(arr, device: torch.device | None = None) -> torch.Tensor:
    """Contiguous device tensor holding `arr` (zero-copy when it already is one)."""
    if isinstance(arr, np.ndarray):
        ti_type(arr)  # KeyError for unsupported dtypes
        dev = device or default_device()
        return torch.from_numpy(np.ascontiguousarray(arr)).to(dev)
    if isinstance(arr, torch.Tensor):
        ti_type(arr)
        if arr.is_cuda and (device is None or arr.device == device):
            return arr.contiguous()
        return arr.contiguous().to(device or default_device())
    raise ValueError(f"Unsupported input type {type(arr)}")


def from_device(dev_tensor: torch.Tensor, like):
    """Hand a device result back in the container kind / device of `like`."""
    if isinstance(like, np.ndarray):
        return dev_tensor.cpu().numpy()
    if isinstance(like, torch.Tensor) and like.device != dev_tensor.device:
        return dev_tensor.to(like.device)
    return dev_tensor
