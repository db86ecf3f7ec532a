"""The reference's own workload (bench/camera_isp.py:19-45): Camera16, 6 cameras,
load_packed12 x6 + tonemap_reinhard(gamma=0.6), 4096x3072 packed-12 frames."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import taichi_image_amd as ti
from taichi_image_amd import synthetic
dev = torch.device("cuda", 0)
frames = [torch.from_numpy(synthetic.synthetic_packed12(k % 2)).to(dev) for k in range(6)]
for rw in (0, 1920):
    isp = ti.Camera16(ti.BayerPattern.RGGB, moving_alpha=0.1, resize_width=rw, device=dev)
    def step():
        imgs = [isp.load_packed12(f) for f in frames]
        return isp.tonemap_reinhard(imgs, gamma=0.6)
    for _ in range(5): step()
    torch.cuda.synchronize()
    n = 30
    t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"resize_width={rw}: {dt*1e3:.3f} ms per 6-camera step = {6 * 12.582912 / dt:.0f} MP/s ({dt/6*1e6:.1f} us per frame)")
    # split: load only
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): imgs = [isp.load_packed12(f) for f in frames]
    torch.cuda.synchronize(); dl = (time.perf_counter() - t0) / n
    print(f"   load_packed12 x6: {dl*1e3:.3f} ms ({dl/6*1e6:.1f} us per frame)")
