"""Bounds (FP_LO, FP_HI) of the demosaiced image for the synthetic frames: are they exactly (0, 1)?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from taichi_image_amd import _native, synthetic
from taichi_image_amd.pipeline import pipeline12_reinhard
dev = torch.device("cuda", 0)
for k in range(6):
    frame = torch.from_numpy(synthetic.synthetic_packed12(k)).to(dev)
    out = pipeline12_reinhard(frame, whole_frame=False)
    torch.cuda.synchronize()
    ws = _native.workspace(3072, 4096, dev)
    fp = ws.cpu().numpy().view(np.float32)[:20]
    print(k, "lo", fp[0], "hi", fp[1], "inv", fp[2])
