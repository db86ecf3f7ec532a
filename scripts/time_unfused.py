"""Times the unfused GPU chain at 4K: load_packed (unpack+demosaic -> f16 RGB) and the stateless
tonemap_reinhard on that image (4 elementwise passes)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import taichi_image_amd as ti
from taichi_image_amd import _native, synthetic, types
dev = torch.device("cuda", 0)
H, W = 3072, 4096
frame = torch.from_numpy(synthetic.synthetic_packed12(0)).to(dev)
isp = ti.Camera16(ti.BayerPattern.RGGB, device=dev)
def timeit(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
rgb = isp.load_packed12(frame)
print("load_packed12 (unpack+demosaic -> f16 RGB): %.1f us" % timeit(lambda: isp.load_packed12(frame)))
print("tonemap_reinhard f16->f16 (4 passes + 3 finalize): %.1f us" % timeit(lambda: ti.tonemap.tonemap_reinhard(rgb, dtype=types.f16)))
print("tonemap_linear f16->f16 (2 passes): %.1f us" % timeit(lambda: ti.tonemap.tonemap_linear(rgb, dtype=types.f16)))
