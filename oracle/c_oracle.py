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


# ---- camera_isp.py stateful path (full-size checks of the ISP metering / Reinhard / linear kernels) ----
def _as_f32_images(images):
    ims = [np.ascontiguousarray(im, np.float32) for im in images]
    ptrs = (ctypes.c_void_p * len(ims))(*[im.ctypes.data for im in ims])
    return ims, ptrs


def metering_images(images, alpha, prev, stride=8):
    """camera_isp.py:142-175; images: list of (H, W, 3) f16/f32 arrays; prev: f32[9]."""
    ims, ptrs = _as_f32_images(images)
    H, W, _ = ims[0].shape
    prev = np.ascontiguousarray(prev, np.float32)
    out = np.empty(9, np.float32)
    lib().orc_metering_images(ptrs, len(ims), H, W, int(stride), ctypes.c_float(alpha), _p(prev), _p(out))
    return out


class IspState:
    """The rolling state of camera_isp.ISP (camera_isp.py:267,376-385) on the C oracle."""

    def __init__(self, moving_alpha=0.1, stride=8):
        self.metrics = None
        self.moving_alpha = moving_alpha
        self.stride = stride

    def update_metering(self, images):
        if self.metrics is None:
            self.metrics = metering_images(images, 0.0, np.zeros(9, np.float32), self.stride)
        else:
            self.metrics = metering_images(images, 1.0 - self.moving_alpha, self.metrics, self.stride)
        return self.metrics


def reinhard_isp(image, m, gamma=1.0, intensity=1.0, light_adapt=1.0, color_adapt=0.0):
    """camera_isp.py:177-218 -> (u8 output, image after the in-place write-back, in the image dtype)."""
    dt = image.dtype
    work = np.ascontiguousarray(image, np.float32).copy()
    H, W, _ = work.shape
    out = np.empty((H, W, 3), np.uint8)
    m = np.ascontiguousarray(m, np.float32)
    lib().orc_reinhard_isp(_p(work), H, W, int(dt == np.float16), _p(m), ctypes.c_float(gamma),
                           ctypes.c_float(intensity), ctypes.c_float(light_adapt), ctypes.c_float(color_adapt),
                           _p(out), None)
    return out, work.astype(dt)


def linear_isp(image, m, gamma=1.0):
    src = np.ascontiguousarray(image, np.float32)
    H, W, _ = src.shape
    out = np.empty((H, W, 3), np.uint8)
    m = np.ascontiguousarray(m, np.float32)
    lib().orc_linear_isp(_p(src), H, W, _p(m), ctypes.c_float(gamma), _p(out))
    return out
