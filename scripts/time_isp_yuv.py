"""Camera16 (full resolution, 6 cameras): tonemap_reinhard + rgb_yuv420_image vs the fused tonemap_reinhard_yuv420."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import taichi_image_amd as ti
from taichi_image_amd import synthetic, color
dev = torch.device("cuda", 0)
frames = [torch.from_numpy(synthetic.synthetic_packed12(k % 2)).to(dev) for k in range(6)]
for rw in (0, 1920):
    isp = ti.Camera16(ti.BayerPattern.RGGB, moving_alpha=0.1, resize_width=rw, device=dev)
    def unfused():
        imgs = [isp.load_packed12(f) for f in frames]
        return [color.rgb_yuv420_image(o) for o in isp.tonemap_reinhard(imgs, gamma=0.6)]
    def fused():
        imgs = [isp.load_packed12(f) for f in frames]
        return isp.tonemap_reinhard_yuv420(imgs, gamma=0.6)
    for name, fn in (("tonemap_reinhard + rgb_yuv420_image", unfused), ("tonemap_reinhard_yuv420 (fused)", fused)):
        for _ in range(5): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print(f"resize_width={rw:4d} {name:40s}: {dt*1e3:.3f} ms per 6-camera step ({dt/6*1e6:.1f} us per frame, {6 * 12.582912 / dt:.0f} MP/s)")
