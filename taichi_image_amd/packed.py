"""12/16-bit packed RAW unpack / pack -- call surface of taichi_image/packed.py.

decode12 / decode16 / encode12 keep the reference signatures (packed.py:176-210); the kernels
are HIP (csrc/isp_elementwise.hip: decode12_kernel, decode16_kernel, encode12_kernel).
"""
from __future__ import annotations

import numpy as np
import torch

from . import _native, types
from .types import as_dtype


def _flat_dev(values):
    dev = types.to_device(values)
    return dev.reshape(-1)


def encode12(values, scaled=False, ids_format=False):
    """packed.py:176-185."""
    shape = tuple(values.shape)
    assert shape[-1] % 2 == 0, f"last dimension must be even for 12-bit encoding got: {shape}"
    in_dtype = types.ti_type(values)
    flat = _flat_dev(values)
    n = flat.shape[0]
    enc = torch.empty((n * 3) // 2, dtype=torch.uint8, device=flat.device)
    _native.check(_native.lib().mi_isp_encode12(flat.data_ptr(), enc.data_ptr(), n, in_dtype.code, int(bool(scaled)),
                                                int(bool(ids_format)), _native.stream_ptr(flat.device)))
    return types.from_device(enc.reshape(shape[:-1] + (shape[-1] * 3 // 2,)), values)


def decode12(values, dtype=types.u16, scaled=False, ids_format=False):
    """packed.py:188-198."""
    shape = tuple(values.shape)
    assert types.ti_type(values) == types.u8
    assert shape[-1] % 3 == 0, f"last dimension must be a factor of 3 for 12-bit decoding got: {shape}"
    dtype = as_dtype(dtype)
    flat = _flat_dev(values)
    n_px = (flat.shape[0] * 2) // 3
    out = torch.empty(n_px, dtype=dtype.torch, device=flat.device)
    _native.check(_native.lib().mi_isp_decode12(flat.data_ptr(), out.data_ptr(), n_px, dtype.code, int(bool(scaled)),
                                                int(bool(ids_format)), _native.stream_ptr(flat.device)))
    return types.from_device(out.reshape(shape[:-1] + (shape[-1] * 2 // 3,)), values)


def decode16(values, dtype=types.u16, scaled=False, ids_format=False):
    """packed.py:200-210.  (The reference wrapper forwards `ids_format` to a kernel factory that
    does not take it and raises TypeError; the 16-bit layout has no IDS variant, so the flag is
    accepted and ignored here.)"""
    shape = tuple(values.shape)
    assert types.ti_type(values) == types.u8
    assert shape[-1] % 2 == 0, f"last dimension must be a factor of 2 for 16-bit decoding got: {shape}"
    dtype = as_dtype(dtype)
    flat = _flat_dev(values)
    n_px = flat.shape[0] // 2
    out = torch.empty(n_px, dtype=dtype.torch, device=flat.device)
    _native.check(_native.lib().mi_isp_decode16(flat.data_ptr(), out.data_ptr(), n_px, dtype.code, int(bool(scaled)),
                                                _native.stream_ptr(flat.device)))
    return types.from_device(out.reshape(shape[:-1] + (shape[-1] // 2,)), values)


# Kernel-factory spellings used by camera_isp.py:76,335 -- k(encoded_flat, out_flat) in place.
def decode12_kernel(out_type, scaled=False, ids_format=False):
    dtype = as_dtype(out_type)

    def k(encoded: torch.Tensor, out: torch.Tensor):
        _native.check(_native.lib().mi_isp_decode12(encoded.data_ptr(), out.data_ptr(), out.numel(), dtype.code,
                                                    int(bool(scaled)), int(bool(ids_format)),
                                                    _native.stream_ptr(out.device)))
    return k


def decode16_kernel(out_type, scaled=False):
    dtype = as_dtype(out_type)

    def k(encoded: torch.Tensor, out: torch.Tensor):
        _native.check(_native.lib().mi_isp_decode16(encoded.data_ptr(), out.data_ptr(), out.numel(), dtype.code,
                                                    int(bool(scaled)), _native.stream_ptr(out.device)))
    return k
