"""Driver for rocprofv3: process_packed12 on 6 full-resolution cameras (keep_images from argv[1], default 0)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import taichi_image_amd as ti
from taichi_image_amd import synthetic
dev = torch.device("cuda", 0)
keep = bool(int(sys.argv[1])) if len(sys.argv) > 1 else False
a = ti.Camera16(ti.BayerPattern.RGGB, device=dev)
fr = [torch.from_numpy(synthetic.synthetic_packed12(i)).to(dev) for i in range(6)]
for _ in range(30): a.process_packed12(fr, gamma=0.6, keep_images=keep)
torch.cuda.synchronize()
