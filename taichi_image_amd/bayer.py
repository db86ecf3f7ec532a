"""Bayer mosaic / demosaic -- call surface of taichi_image/bayer.py.

bayer_to_rgb is the LDS-tiled 13-tap diamond demosaic of csrc/isp_tile.h (bayer.py:115-190).
"""
from __future__ import annotations

import enum
from typing import Optional

import numpy as np
import torch

from . import _native, types
from .types import as_dtype


class BayerPattern(enum.Enum):
    """bayer.py:75-83."""
    RGGB = 0
    GRBG = 1
    GBRG = 2
    BGGR = 3

    @property
    def pixel_order(self):
        return pixel_orders[self]


pixel_orders = {  # bayer.py:85-90
    BayerPattern.RGGB: (0, 1, 1, 2),
    BayerPattern.GRBG: (1, 0, 2, 1),
    BayerPattern.GBRG: (1, 2, 0, 1),
    BayerPattern.BGGR: (2, 1, 1, 0),
}

kernel_patterns = {  # bayer.py:92-97
    BayerPattern.RGGB: (0, 1, 2, 3),
    BayerPattern.GBRG: (1, 0, 3, 2),
    BayerPattern.GRBG: (2, 3, 0, 1),
    BayerPattern.BGGR: (3, 2, 1, 0),
}


def bayer_weights() -> np.ndarray:
    """The [4][13][3] integer weight tables compiled into the HIP kernels (bayer.py:30-55)."""
    import ctypes
    buf = (ctypes.c_int32 * (4 * 13 * 3))()
    _native.check(_native.lib().mi_isp_bayer_weights(buf))
    return np.array(buf, dtype=np.int32).reshape(4, 13, 3)


def rgb_to_bayer(image, pattern: BayerPattern = BayerPattern.RGGB):
    """bayer.py:193-198."""
    assert image.ndim == 3 and image.shape[2] == 3, "image must be RGB"
    dtype = types.ti_type(image)
    dev = types.to_device(image)
    H, W = dev.shape[:2]
    cfa = torch.empty((H, W), dtype=dtype.torch, device=dev.device)
    _native.check(_native.lib().mi_isp_mosaic(dev.data_ptr(), cfa.data_ptr(), H, W, dtype.code, pattern.value,
                                              _native.stream_ptr(dev.device)))
    return types.from_device(cfa, image)


def bayer_to_rgb(bayer, pattern: BayerPattern = BayerPattern.RGGB, correct_colors: Optional[np.ndarray] = None,
                 dtype=None):
    """bayer.py:202-219."""
    assert bayer.ndim == 2, "image must be mono bayer"
    assert bayer.shape[0] % 2 == 0 and bayer.shape[1] % 2 == 0, "image must be even size"
    in_dtype = types.ti_type(bayer)
    out_dtype = in_dtype if dtype is None else as_dtype(dtype)
    dev = types.to_device(bayer)
    H, W = dev.shape
    rgb = torch.empty((H, W, 3), dtype=out_dtype.torch, device=dev.device)
    _native.check(_native.lib().mi_isp_demosaic(dev.data_ptr(), rgb.data_ptr(), H, W, in_dtype.code, out_dtype.code,
                                                pattern.value, _native.ccm_arg(correct_colors),
                                                _native.stream_ptr(dev.device)))
    return types.from_device(rgb, bayer)
