"""Stateless per-image tonemaps -- call surface of taichi_image/tonemap.py."""
from __future__ import annotations

import torch

from . import _native, types
from .types import as_dtype


def _check_rgb(dev):
    assert dev.ndim == 3 and dev.shape[2] == 3, "image must be (H, W, 3)"


def tonemap_linear(src, gamma=1.0, dtype=types.u8):
    """tonemap.py:41-46: min/max-normalised gamma curve."""
    in_dtype = types.ti_type(src)
    out_dtype = as_dtype(dtype)
    dev = types.to_device(src)
    _check_rgb(dev)
    H, W = dev.shape[:2]
    out = torch.empty((H, W, 3), dtype=out_dtype.torch, device=dev.device)
    ws = _native.workspace(H, W, dev.device)
    _native.check(_native.lib().mi_isp_tonemap_linear(dev.data_ptr(), out.data_ptr(), H, W, in_dtype.code,
                                                      out_dtype.code, float(gamma), ws.data_ptr(),
                                                      _native.stream_ptr(dev.device)))
    return types.from_device(out, src)


def tonemap_reinhard(src, gamma=1.0, intensity=1.0, light_adapt=1.0, color_adapt=0.0, dtype=types.u8):
    """tonemap.py:160-168: Reinhard-2005 photoreceptor tonemap with in-kernel statistics."""
    in_dtype = types.ti_type(src)
    out_dtype = as_dtype(dtype)
    dev = types.to_device(src)
    _check_rgb(dev)
    H, W = dev.shape[:2]
    out = torch.empty((H, W, 3), dtype=out_dtype.torch, device=dev.device)
    ws = _native.workspace(H, W, dev.device)
    _native.check(_native.lib().mi_isp_tonemap_reinhard(dev.data_ptr(), out.data_ptr(), H, W, in_dtype.code,
                                                        out_dtype.code, float(gamma), float(intensity),
                                                        float(light_adapt), float(color_adapt), ws.data_ptr(),
                                                        _native.stream_ptr(dev.device)))
    return types.from_device(out, src)
