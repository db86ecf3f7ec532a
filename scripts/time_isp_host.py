"""Host-side issue time of the Camera16 step (is the ISP path launch-bound in Python?)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import taichi_image_amd as ti
from taichi_image_amd import synthetic
dev = torch.device("cuda", 0)
frames = [torch.from_numpy(synthetic.synthetic_packed12(k % 2)).to(dev) for k in range(6)]
isp = ti.Camera16(ti.BayerPattern.RGGB, moving_alpha=0.1, resize_width=1920, device=dev)
def step():
    imgs = [isp.load_packed12(f) for f in frames]
    return isp.tonemap_reinhard(imgs, gamma=0.6)
for _ in range(5): step()
torch.cuda.synchronize()
n = 50
t0 = time.perf_counter()
for _ in range(n): step()
t_issue = (time.perf_counter() - t0) / n
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / n
print(f"host issue {t_issue*1e6:.0f} us per step, wall {t_all*1e6:.0f} us per step")
t0 = time.perf_counter()
for _ in range(n): imgs = [isp.load_packed12(f) for f in frames]
t_l = (time.perf_counter() - t0) / n
torch.cuda.synchronize()
print(f"host issue of 6 x load_packed12: {t_l*1e6:.0f} us ({t_l/6*1e6:.1f} us per call)")
