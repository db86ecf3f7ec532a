"""RGB <-> planar YUV 4:2:0 and the grey weights: the reference's `taichi_image.color`
(`color/__init__.py:7-10`, `color/yuv_420.py`), the step after the ISP path.

Same call surface: `rgb_yuv420_image(src, dtype=None)`, `yuv420_rgb_image(yuv, dtype=None)`,
`split_yuv_420(yuv)`; numpy in -> numpy out, torch in -> torch out on the same device.  The
reference's quirks are kept (see csrc/isp_elementwise.hip): the BGR-named matrix is applied to
rgb.bgr, and `tm.clamp(0, 1, x)` clamps from above only.
"""
from __future__ import annotations

import torch

from . import _native, types
from .types import as_dtype

GRAY_WEIGHTS = (0.299, 0.587, 0.114)      # color/__init__.py:7-10 (rgb_gray)


def split_yuv_420(yuv):
    """yuv_420.py:95-103: (Y plane, (2, H/2, W/2) chroma planes, (width, height))."""
    height = yuv.shape[0] * 2 // 3
    width = yuv.shape[1]
    y = yuv[:height]
    uv = yuv[height:].reshape(2, height // 2, width // 2)
    return y, uv, (width, height)


def rgb_yuv420_image(src, dtype=None):
    """yuv_420.py:105-119."""
    assert src.ndim == 3 and src.shape[2] == 3, "image must be RGB"
    in_dtype = types.ti_type(src)
    out_dtype = in_dtype if dtype is None else as_dtype(dtype)
    height, width, _ = src.shape
    assert height % 2 == 0 and width % 2 == 0, "image must be even size"
    dev = types.to_device(src)
    # every element is written (even sizes are asserted), so no zero fill as in the reference (:112)
    yuv = torch.empty(((height * 3) // 2, width), dtype=out_dtype.torch, device=dev.device)
    _native.check(_native.lib().mi_isp_rgb_to_yuv420(dev.data_ptr(), yuv.data_ptr(), height, width, in_dtype.code,
                                                     out_dtype.code, _native.stream_ptr(dev.device)))
    return types.from_device(yuv, src)


def yuv420_rgb_image(yuv, dtype=None):
    """yuv_420.py:121-131."""
    assert yuv.ndim == 2, "yuv image must be 2-D (H * 3 / 2, W)"
    in_dtype = types.ti_type(yuv)
    out_dtype = in_dtype if dtype is None else as_dtype(dtype)
    h, w = yuv.shape[0] * 2 // 3, yuv.shape[1]
    assert h % 2 == 0 and w % 2 == 0 and h * 3 // 2 == yuv.shape[0], "yuv image must hold an even-sized frame"
    dev = types.to_device(yuv)
    rgb = torch.empty((h, w, 3), dtype=out_dtype.torch, device=dev.device)
    _native.check(_native.lib().mi_isp_yuv420_to_rgb(dev.data_ptr(), rgb.data_ptr(), h, w, in_dtype.code,
                                                     out_dtype.code, _native.stream_ptr(dev.device)))
    return types.from_device(rgb, yuv)
