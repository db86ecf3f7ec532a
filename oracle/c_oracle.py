"""ctypes wrapper of oracle/liborc_isp.so (the C/OpenMP restatement).  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "liborc_isp.so")
_lib = None


def available() -> bool:
    return os.path.exists(_PATH)


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(_PATH)
        L.orc_threads.restype = ctypes.c_int
        L.orc_f32_to_f16_bits.restype = ctypes.c_uint16
        L.orc_f32_to_f16_bits.argtypes = [ctypes.c_float]
        _lib = L
    return _lib


def threads() -> int:
    return int(lib().orc_threads())


def set_threads(n: int) -> None:
    lib().orc_set_threads(ctypes.c_int(int(n)))


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def bayer_kernels():
    out = np.zeros(4 * 13 * 3, np.int32)
    lib().orc_bayer_kernels(_p(out))
    return out.reshape(4, 13, 3)


def decode12_scaled(enc, ids=False, work="f16"):
    enc = np.ascontiguousarray(enc, np.uint8).reshape(-1)
    out = np.empty(enc.size * 2 // 3, np.float32)
    lib().orc_decode12_scaled(_p(enc), ctypes.c_int64(enc.size // 3), int(ids), int(work == "f16"), _p(out))
    return out


def demosaic(cfa_f32, pattern=0, in_scale=1.0, ccm=None, round_f16=False):
    cfa = np.ascontiguousarray(cfa_f32, np.float32)
    H, W = cfa.shape
    rgb = np.empty((H, W, 3), np.float32)
    c = None if ccm is None else np.ascontiguousarray(np.asarray(ccm, np.float64).reshape(9).astype(np.float32))
    lib().orc_demosaic(_p(cfa), H, W, int(pattern), ctypes.c_float(in_scale), None if c is None else _p(c),
                       int(round_f16), _p(rgb))
    return rgb


_KIND = {"u8": (0, np.uint8), "f16": (2, np.uint16), "f32": (3, np.float32)}


def pipeline12_reinhard(packed, pattern=0, ids=False, work="f16", out="f16", gamma=1.0, intensity=1.0,
                        light_adapt=1.0, color_adapt=0.0):
    packed = np.ascontiguousarray(packed, np.uint8)
    H, W = packed.shape[0], packed.shape[1] * 2 // 3
    kind, dt = _KIND[out]
    res = np.empty((H, W, 3), dt)
    lib().orc_pipeline12_reinhard(_p(packed), H, W, int(ids), int(pattern), int(work == "f16"), ctypes.c_float(gamma),
                                  ctypes.c_float(intensity), ctypes.c_float(light_adapt), ctypes.c_float(color_adapt),
                                  kind, _p(res))
    return res.view(np.float16) if out == "f16" else res
