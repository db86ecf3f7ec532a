"""ctypes binding of libmi355_isp.so (include/mi_isp.h) and device-buffer plumbing.

PyTorch-ROCm tensors are only the device-buffer / stream provider here.  There is no CPU
fallback: every public op raises if the HIP library or a GPU is missing.
"""
from __future__ import annotations

import ctypes
import os
import threading
from ctypes import c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p, POINTER

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# MI_ISP_LIB: an instrumented build of the same library (measurement only, e.g. scripts/tile_stamps.py)
LIB_PATH = os.environ.get("MI_ISP_LIB") or os.path.join(_HERE, "lib", "libmi355_isp.so")

MI_U8, MI_U16, MI_F16, MI_F32 = 0, 1, 2, 3

# every symbol include/mi_isp.h declares: name -> (restype, argtypes)
_P = c_void_p
SIGNATURES = {
    "mi_isp_version": (c_int, []),
    "mi_isp_last_error": (c_char_p, []),
    "mi_isp_bayer_weights": (c_int, [POINTER(c_int32)]),
    "mi_isp_workspace_bytes": (c_size_t, [c_int, c_int]),
    "mi_isp_decode12": (c_int, [_P, _P, c_int64, c_int, c_int, c_int, _P]),
    "mi_isp_decode16": (c_int, [_P, _P, c_int64, c_int, c_int, _P]),
    "mi_isp_encode12": (c_int, [_P, _P, c_int64, c_int, c_int, c_int, _P]),
    "mi_isp_load_convert": (c_int, [_P, _P, c_int64, c_int, c_int, _P]),
    "mi_isp_demosaic": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, POINTER(c_float), _P]),
    "mi_isp_mosaic": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "mi_isp_rgb_to_yuv420": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "mi_isp_yuv420_to_rgb": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "mi_isp_resize_bilinear": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_float, c_float, c_int, c_int, _P]),
    "mi_isp_transform": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "mi_isp_metering": (c_int, [POINTER(_P), c_int, c_int, c_int, c_int, c_int, _P, c_float, _P, _P]),
    "mi_isp_metering_to": (c_int, [POINTER(_P), c_int, c_int, c_int, c_int, c_int, _P, _P, c_float, _P, _P]),
    "mi_isp_metering_bounds": (c_int, [POINTER(_P), c_int, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "mi_isp_metering_sums": (c_int, [POINTER(_P), c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P]),
    "mi_isp_reinhard": (c_int, [_P, _P, c_int, c_int, c_int, _P, c_float, c_float, c_float, c_float, c_int, _P, _P]),
    "mi_isp_reinhard_batch": (c_int, [POINTER(_P), POINTER(_P), c_int, c_int, c_int, c_int, _P, c_float, c_float, c_float,
                                      c_float, c_int, _P, _P]),
    "mi_isp_reinhard_batch_keep": (c_int, [POINTER(_P), POINTER(_P), c_int, c_int, c_int, c_int, _P, c_float, c_float, c_float,
                                           c_float, c_int, _P, _P]),
    "mi_isp_reinhard_batch_yuv420": (c_int, [POINTER(_P), POINTER(_P), c_int, c_int, c_int, c_int, _P, c_float, c_float,
                                             c_float, c_float, _P, _P]),
    "mi_isp_linear_batch": (c_int, [POINTER(_P), POINTER(_P), c_int, c_int, c_int, c_int, _P, c_float, c_int, _P, _P]),
    "mi_isp_linear": (c_int, [_P, _P, c_int, c_int, c_int, _P, c_float, c_int, _P, _P]),
    "mi_isp_tonemap_linear": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_float, _P, _P]),
    "mi_isp_tonemap_reinhard": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_float, c_float, c_float, c_float, _P, _P]),
    "mi_isp_load_packed": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, POINTER(c_float), c_int, c_int, c_int,
                                   c_float, _P]),
    "mi_isp_load_packed_metered": (c_int, [_P, _P, c_int, c_int, c_int, c_int, c_int, POINTER(c_float), c_int, c_int, c_int,
                                           c_float, _P, c_int, _P]),
    "mi_isp_load_packed_batch": (c_int, [POINTER(_P), POINTER(_P), POINTER(_P), c_int, c_int, c_int, c_int, c_int, c_int,
                                         POINTER(c_float), c_int, c_int, c_int, c_float, c_int, _P]),
    "mi_isp_load_packed_metered_is_fused": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "mi_isp_load_packed_scale_supported": (c_int, [c_float]),
    "mi_isp_pipeline12_reinhard": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, POINTER(c_float), c_int, c_int,
                                           c_float, c_float, c_float, c_float, _P, _P]),
    "mi_isp_pipeline12_reinhard_batch": (c_int, [POINTER(_P), POINTER(_P), POINTER(_P), c_int, c_int, c_int, c_int, c_int,
                                                 POINTER(c_float), c_int, c_int, c_float, c_float, c_float, c_float,
                                                 _P, POINTER(_P), c_int]),
    "mi_isp_pipeline12_pass": (c_int, [_P, _P, c_int, c_int, c_int, c_int, POINTER(c_float), c_int, c_int, c_float,
                                       c_float, c_float, c_int, _P, _P]),
    "mi_isp_metering_combine_bounds": (c_int, [_P, c_int, _P, c_float, _P, _P]),
    "mi_isp_metering_combine_sums": (c_int, [_P, c_int, _P, _P, c_float, _P]),
    "mi_isp_camera_frame_batch": (c_int, [POINTER(_P), POINTER(_P), POINTER(_P), c_int, c_int, c_int, c_int, c_int, c_int,
                                          POINTER(c_float), c_int, c_int, c_int, c_float, c_int, _P, c_float, c_int, c_float,
                                          c_float, c_float, c_float, c_int, _P, _P]),
    "mi_isp_camera_group_reinhard": (c_int, [POINTER(_P), POINTER(_P), POINTER(_P), c_int, c_int, c_int, c_int, POINTER(c_float),
                                             _P, _P, c_float, c_float, c_float, c_float, c_float, _P, _P, _P]),
    "mi_isp_camera_group_subsample": (c_int, [POINTER(_P), c_int, c_int, c_int, c_int, POINTER(c_float), _P, _P]),
    "mi_isp_camera_group_tonemap": (c_int, [POINTER(_P), POINTER(_P), POINTER(_P), c_int, c_int, c_int, c_int, POINTER(c_float),
                                            _P, c_float, c_float, c_float, c_float, _P, _P]),
    "mi_isp_camera_group_fits": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "mi_isp_camera_group_scratch_bytes": (ctypes.c_size_t, [c_int, c_int, c_int]),
    "mi_isp_camera_group_faults": (c_int, [c_int]),
    "mi_isp_camera_group_set_poll_limit": (c_int, [ctypes.c_uint]),
    "mi_isp_pipeline12_graph_create": (c_int, [POINTER(_P), POINTER(_P), POINTER(_P), c_int, c_int, c_int, c_int, c_int,
                                               POINTER(c_float), c_int, c_int, c_float, c_float, c_float, c_float, _P,
                                               c_int, c_int, POINTER(_P)]),
    "mi_isp_pipeline12_graph_launch": (c_int, [_P, _P]),
    "mi_isp_pipeline12_graph_destroy": (c_int, [_P]),
    "mi_isp_pipeline12_reinhard_whole_frame": (c_int, [_P, _P, c_int, c_int, c_int, c_int, POINTER(c_float), c_int, c_float, c_float,
                                               c_float, c_float, _P, _P]),
    "mi_isp_pipeline12_whole_frame_fits": (c_int, [c_int, c_int, c_int]),
    "mi_isp_pipeline12_reinhard_whole_frame_batch": (c_int, [POINTER(_P), POINTER(_P), c_int, c_int, c_int, c_int, c_int,
                                                             POINTER(c_float), c_int, c_float, c_float, c_float, c_float, _P, _P]),
    "mi_isp_workspace_check": (c_int, [_P, c_int, c_int, c_int, POINTER(c_int), POINTER(c_int), _P]),
    "mi_isp_whole_frame_faults": (c_int, [c_int]),
    "mi_isp_whole_frame_set_poll_limit": (c_int, [ctypes.c_uint]),
    "mi_isp_whole_frame_set_sabotage": (c_int, [c_int]),
    "mi_isp_metering_faults": (c_int, [c_int]),
    "mi_isp_metering_set_poll_limit": (c_int, [ctypes.c_uint]),
    "mi_isp_reinhard_faults": (c_int, [c_int]),
    "mi_isp_reinhard_set_poll_limit": (c_int, [ctypes.c_uint]),
    "mi_isp_workspace_error_offset": (ctypes.c_size_t, [c_int, c_int]),
    "mi_isp_profile_enable": (c_int, [c_int, c_int]),
    "mi_isp_profile_collect": (c_int, [POINTER(c_float), POINTER(c_int)]),   # float[4]
}

_lib = None
_lock = threading.Lock()


class NativeLibraryError(ImportError):
    pass


def lib() -> ctypes.CDLL:
    """Load the HIP library (once).  Raises loudly when it has not been built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise NativeLibraryError(
                        f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                        "(or `make -C taichi_image_amd/csrc`).  There is no CPU fallback.")
                L = ctypes.CDLL(LIB_PATH)
                guarded = _GuardedLibrary()
                for name, (res, args) in SIGNATURES.items():
                    fn = getattr(L, name)   # AttributeError if the library lacks a declared symbol
                    fn.restype, fn.argtypes = res, args
                    setattr(guarded, name, _Guarded(fn))
                _lib = guarded
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        raise RuntimeError("libmi355_isp: " + lib().mi_isp_last_error().decode("utf-8", "replace"))


def require_gpu() -> None:
    if not torch.cuda.is_available():
        raise RuntimeError("taichi_image_amd needs an MI355X (HIP) device; there is no CPU fallback")


class _StreamArg(int):
    """A hipStream_t value that remembers which device it belongs to (see _Guarded)."""
    device = None


def stream_ptr(device: torch.device) -> int:
    s = _StreamArg(torch.cuda.current_stream(device).cuda_stream)
    s.device = device
    return s


class _Guarded:
    """A library entry point that launches on the device its stream argument belongs to.

    The C ABI launches on the CURRENT device (a null stream is every device's default stream), while the call surface
    takes `device=` arguments and tensors of any GPU (camera_isp.py:239-251 of the reference does too).  Every wrapper
    passes its stream through stream_ptr(device); when that device is not the current one the call runs under
    torch.cuda.device(device), so pointers, workspace and launch agree."""

    def __init__(self, fn):
        self.fn = fn

    def __call__(self, *args):
        dev = next((a.device for a in args if isinstance(a, _StreamArg)), None)
        if dev is None or dev.type != "cuda" or dev.index is None or dev.index == torch.cuda.current_device():
            return self.fn(*args)
        with torch.cuda.device(dev):
            return self.fn(*args)


class _GuardedLibrary:
    pass


def ccm_arg(correct_colors):
    """3x3 colour matrix (row-major) -> float[9] or NULL (bayer.py:210-211)."""
    if correct_colors is None:
        return None
    m = np.asarray(correct_colors, dtype=np.float64).reshape(-1)
    assert m.size == 9, "colour correction must be a 3x3 matrix"
    return (c_float * 9)(*[float(v) for v in m])


_ws_cache: dict = {}


def workspace(H: int, W: int, device: torch.device, slots: int = 1) -> torch.Tensor:
    """Scratch buffer for one in-flight call on (device, current stream)."""
    nbytes = int(lib().mi_isp_workspace_bytes(int(H), int(W))) * slots
    key = (device.index or 0, stream_ptr(device), nbytes)
    ws = _ws_cache.get(key)
    if ws is None:
        if len(_ws_cache) > 64:
            _ws_cache.clear()
        ws = torch.zeros(nbytes, dtype=torch.uint8, device=device)   # arrival counters start at 0
        _ws_cache[key] = ws
    return ws




def ptr_array(tensors):
    arr = (c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])
    return arr
