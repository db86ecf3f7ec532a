"""taichi_image_amd -- MI355X-native camera-ISP hot path behind taichi_image's call surface.

    from taichi_image_amd import camera_isp, bayer, packed, tonemap, interpolate

mirror the modules of uc-vision/taichi_image; every op dispatches through ctypes into
libmi355_isp.so (hand-written HIP for gfx950).  There is no CPU fallback.
"""
from . import types  # noqa: F401
from . import packed, bayer, interpolate, tonemap, camera_isp, pipeline, distributed, color, ingest  # noqa: F401
from .bayer import BayerPattern  # noqa: F401
from .interpolate import ImageTransform  # noqa: F401
from .camera_isp import Camera16, Camera32  # noqa: F401

__version__ = "0.1.0"
