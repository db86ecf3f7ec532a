"""Times the YUV 4:2:0 conversions on a 4K frame (events on the stream)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from taichi_image_amd import color
H, W = 3072, 4096
dev = torch.device("cuda", 0)
def t(fn, reps=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for name, dt, odt, bpp in (("u8 -> u8", torch.uint8, "u8", 3 + 1.5), ("f16 -> u8", torch.float16, "u8", 6 + 1.5), ("f16 -> f16", torch.float16, "f16", 6 + 3)):
    img = (torch.rand((H, W, 3), device=dev) * (255 if dt == torch.uint8 else 1)).to(dt)
    us = t(lambda: color.rgb_yuv420_image(img, dtype=odt))
    print(f"rgb_yuv420 {name}: {us:7.1f} us  ({H * W * bpp / us / 1e6:5.2f} TB/s algorithmic, includes the output memset + alloc)")
    yuv = color.rgb_yuv420_image(img, dtype=odt)
    us = t(lambda: color.yuv420_rgb_image(yuv))
    print(f"yuv420_rgb {odt} -> {odt}: {us:7.1f} us")
