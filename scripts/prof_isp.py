"""Profiling driver for the ISP path: Camera16(resize_width=1920) load_packed12 x n (+ tonemap)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import taichi_image_amd as ti
from taichi_image_amd import synthetic
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
frame = torch.from_numpy(synthetic.synthetic_packed12(0)).to(dev)
isp = ti.Camera16(ti.BayerPattern.RGGB, moving_alpha=0.1, resize_width=1920, device=dev)
for _ in range(n):
    imgs = [isp.load_packed12(frame)]
out = isp.tonemap_reinhard(imgs, gamma=0.6)
torch.cuda.synchronize()
print("done")
