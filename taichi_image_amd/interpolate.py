"""Bilinear resize and orientation transforms -- call surface of taichi_image/interpolate.py."""
from __future__ import annotations

from enum import Enum

import numpy as np
import torch

from . import _native, types
from .types import as_dtype


class ImageTransform(Enum):
    """interpolate.py:9-17."""
    none = 'none'
    rotate_90 = 'rotate_90'
    rotate_180 = 'rotate_180'
    rotate_270 = 'rotate_270'
    transpose = 'transpose'
    flip_horiz = 'flip_horiz'
    flip_vert = 'flip_vert'
    transverse = 'transverse'


_TRANSFORM_CODE = {t: i for i, t in enumerate(ImageTransform)}


def transform_code(t: ImageTransform) -> int:
    return _TRANSFORM_CODE[t]


def transformed_size(size, transform: ImageTransform):
    """interpolate.py:112-117 (note: `transverse` is not in the swap list in the reference)."""
    w, h = size
    if transform in [ImageTransform.rotate_90, ImageTransform.rotate_270, ImageTransform.transpose]:
        return (h, w)
    return (w, h)


def transform(src, transform: ImageTransform):
    """interpolate.py:119-125.  `transverse` is only defined for square images: the reference
    keeps the source shape for it while its index map needs swapped dims, i.e. it reads out of
    bounds on non-square input; that case raises here instead."""
    dtype = types.ti_type(src)
    dev = types.to_device(src)
    Hs, Ws = dev.shape[:2]
    assert dev.ndim == 3 and dev.shape[2] == 3, "image must be (H, W, 3)"
    assert transform != ImageTransform.transverse or Hs == Ws, "transverse is only defined for square images"
    size = transformed_size((Hs, Ws), transform)
    dst = torch.empty((size[0], size[1], 3), dtype=dtype.torch, device=dev.device)
    _native.check(_native.lib().mi_isp_transform(dev.data_ptr(), dst.data_ptr(), Hs, Ws, dtype.code,
                                                 transform_code(transform), _native.stream_ptr(dev.device)))
    return types.from_device(dst, src)


def _scale2(scale):
    if np.isscalar(scale):
        return float(scale), float(scale)
    s = tuple(float(v) for v in scale)
    assert len(s) == 2, "scale must be a scalar or a (row, col) pair"
    return s


def resize_bilinear(src, size, scale=None, dtype=None):
    """interpolate.py:128-139.  size = (w, h).  scale: scalar or (row_scale, col_scale).
    scale=None reproduces the reference exactly, including its crossed axes
    (vec2(size=(w,h)) / vec2(src.shape=(H,W)), interpolate.py:132-133) -- pass an explicit
    scale for a geometrically correct resize, as camera_isp.ISP always does."""
    in_dtype = types.ti_type(src)
    out_dtype = in_dtype if dtype is None else as_dtype(dtype)
    dev = types.to_device(src)
    assert dev.ndim == 3 and dev.shape[2] == 3, "image must be (H, W, 3)"
    Hs, Ws = dev.shape[:2]
    Wd, Hd = int(size[0]), int(size[1])
    if scale is None:
        scale = (Wd / Hs, Hd / Ws)
    s0, s1 = _scale2(scale)
    dst = torch.empty((Hd, Wd, 3), dtype=out_dtype.torch, device=dev.device)
    _native.check(_native.lib().mi_isp_resize_bilinear(dev.data_ptr(), dst.data_ptr(), Hs, Ws, Hd, Wd, s0, s1,
                                                       in_dtype.code, out_dtype.code,
                                                       _native.stream_ptr(dev.device)))
    return types.from_device(dst, src)


def resize_width(src, width: int, dtype=None):
    """interpolate.py:141-145 (height truncated with int())."""
    h, w = src.shape[:2]
    scale = width / w
    return resize_bilinear(src, (width, int(h * scale)), scale, dtype)


def scale_bilinear(src, scale, dtype=None):
    """interpolate.py:147-151: size = ivec2(vec2(w, h) * scale) (f32 product, truncated)."""
    h, w = src.shape[:2]
    sw, sh = _scale2(scale)
    size = (int(np.float32(w) * np.float32(sw)), int(np.float32(h) * np.float32(sh)))
    return resize_bilinear(src, size, scale, dtype=dtype)
