"""Dump the tagged barrier records the whole-frame kernel left in the workspace (debug aid)."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from taichi_image_amd import _native, synthetic
from taichi_image_amd.pipeline import pipeline12_reinhard
dev = torch.device("cuda", 0)
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 512)
frame = torch.from_numpy(synthetic.synthetic_packed12(1, H, W)).to(dev)
out = pipeline12_reinhard(frame, whole_frame=True)
torch.cuda.synchronize()
ws = _native.workspace(H, W, dev).cpu().numpy()
u = ws.view(np.uint32); f = ws.view(np.float32)
cap = 4096
print("epoch", u[60], "error", u[62], "FP[0:19]", f[:19])
for name, row, nch in (("bar0", 20, 3), ("bar1", 28, 3), ("bar2", 36, 1)):
    base = 64 + row * cap
    for b in range(min(4, int(sys.argv[3]) if len(sys.argv) > 3 else 2)):
        o = base + b * 64
        print(name, "block", b, [(f[o + 4 * c: o + 4 * c + 3].tolist(), int(u[o + 4 * c + 3])) for c in range(nch)])
