"""Profiling driver: ISP.load_packed12 with resize_width=1920 on a 4K frame, repeated (for rocprofv3 --pmc).
MI_ISP_NO_STREAM=1 with a -DMI_ISP_MEASURE build selects the round-1 tile kernel (rtile::resize_tile_kernel)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import taichi_image_amd as ti
from taichi_image_amd import synthetic
dev = torch.device("cuda", 0)
frame = torch.from_numpy(synthetic.synthetic_packed12(0)).to(dev)
isp = ti.Camera16(ti.BayerPattern.RGGB, resize_width=1920, device=dev)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    isp.load_packed12(frame)
torch.cuda.synchronize()
print("done")
