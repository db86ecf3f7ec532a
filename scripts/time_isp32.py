"""Camera32 variant of the reference's canonical sequence (test/camera_isp.py:29-39): f32 work dtype."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import taichi_image_amd as ti
from taichi_image_amd import synthetic
dev = torch.device("cuda", 0)
frames = [torch.from_numpy(synthetic.synthetic_packed12(k % 2)).to(dev) for k in range(6)]
for cam, rw in (("Camera32", 1280), ("Camera32", 0), ("Camera16", 1280)):
    isp = getattr(ti, cam)(ti.BayerPattern.RGGB, moving_alpha=1.0, resize_width=rw, device=dev)
    def step():
        imgs = [isp.load_packed12(f) for f in frames]
        return isp.tonemap_reinhard(imgs, gamma=0.6)
    for _ in range(5): step()
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): imgs = [isp.load_packed12(f) for f in frames]
    torch.cuda.synchronize(); dl = (time.perf_counter() - t0) / n
    print(f"{cam} resize_width={rw}: {dt*1e3:.3f} ms per 6-camera step = {6 * 12.582912 / dt:.0f} MP/s ({dt/6*1e6:.1f} us per frame; load {dl/6*1e6:.1f} us per frame)")
