"""Times the standalone operators of the reference's API at their BASELINE sizes."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import taichi_image_amd as ti
from taichi_image_amd import bayer, packed, tonemap, interpolate, synthetic, types
dev = torch.device("cuda", 0)
def t(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
rng = np.random.default_rng(0)
for name, H, W, dt in (("config 1: u16 1080p", 1080, 1920, torch.uint16), ("u16 4K", 3072, 4096, torch.uint16), ("u8 4K", 3072, 4096, torch.uint8),
                       ("f16 4K", 3072, 4096, torch.float16), ("f32 4K", 3072, 4096, torch.float32)):
    if dt in (torch.uint16, torch.uint8):
        cfa = torch.from_numpy(rng.integers(0, 65536 if dt == torch.uint16 else 256, (H, W)).astype(np.uint16 if dt == torch.uint16 else np.uint8)).to(dev)
    else:
        cfa = torch.rand((H, W), device=dev).to(dt)
    us = t(lambda: bayer.bayer_to_rgb(cfa))
    print(f"bayer_to_rgb {name:20s}: {us:8.1f} us  {H * W / us:9.0f} MP/s")
frame = torch.from_numpy(synthetic.synthetic_packed12(0)).to(dev)
us = t(lambda: packed.decode12(frame, dtype=types.f16, scaled=True)); print(f"decode12 4K -> f16 scaled         : {us:8.1f} us")
rgb = torch.rand((3072, 4096, 3), device=dev).to(torch.float16)
us = t(lambda: tonemap.tonemap_reinhard(rgb, dtype=types.f16)); print(f"tonemap_reinhard f16 4K -> f16     : {us:8.1f} us")
us = t(lambda: tonemap.tonemap_linear(rgb, dtype=types.u8)); print(f"tonemap_linear f16 4K -> u8        : {us:8.1f} us")
us = t(lambda: interpolate.resize_width(rgb, 1920)); print(f"resize_width f16 4K -> 1920        : {us:8.1f} us")
from taichi_image_amd.pipeline import pipeline12_reinhard
for odt in ("f16", "u8", "u16", "f32"):
    us = t(lambda: pipeline12_reinhard(frame, dtype=getattr(types, odt), whole_frame=False))
    print(f"pipeline12_reinhard f16 work -> {odt:3s}          : {us:8.1f} us  {12.582912 / us * 1e6:9.0f} MP/s")
