"""Phase timeline of the whole-frame kernel from in-kernel s_memtime stamps (lane 0 of every wave).
    make -C taichi_image_amd/csrc EXTRA="-DMI_STREAM_STAMPS -DMI_ISP_MEASURE" OBJDIR=../../build/csrc_stamps OUT=../lib/libmi355_isp_stamps.so
    MI_ISP_LIB=taichi_image_amd/lib/libmi355_isp_stamps.so python scripts/mega_stamps.py
"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from taichi_image_amd import _native, synthetic
from taichi_image_amd.pipeline import pipeline12_reinhard
H, W = 3072, 4096
dev = torch.device("cuda", 0)
frame = torch.from_numpy(synthetic.synthetic_packed12(0)).to(dev)
out = torch.empty((H, W, 3), dtype=torch.float16, device=dev)
for _ in range(5): pipeline12_reinhard(frame, out=out, whole_frame=True)
torch.cuda.synchronize()
ws = _native.workspace(H, W, dev)
ws[(64 + 20 * 4096) * 4:].zero_()            # the stamp area only (the sync words must stay as the kernel left them)
pipeline12_reinhard(frame, out=out, whole_frame=True)
torch.cuda.synchronize()
raw = ws.cpu().numpy().view(np.uint32)
base = 64 + 20 * 4096
nw = 2048
s = raw[base:base + nw * 16].reshape(nw, 16).astype(np.int64)
names = ["entry", "phase A done", "reduced+signalled", "barrier 0 + bounds", "stats folded", "phase C done", "reduced+signalled",
         "barrier 2 + bounds2", "phase D done"]
rel = (s - s[:, :1]) & 0xFFFFFFFF
print(f"{'phase':24s}   mean    p10    p50    p90   [ticks since the previous stamp]")
for i in range(1, 9):
    d = rel[:, i] - rel[:, i - 1]
    print(f"{names[i]:24s} {d.mean():7.0f} {np.percentile(d,10):6.0f} {np.percentile(d,50):6.0f} {np.percentile(d,90):6.0f}")
print("wave lifetime", rel[:, 8].mean(), np.percentile(rel[:, 8], [10, 50, 90]))

# the folding wave of each block: stamps 9..11 (barrier 0) and 12..14 (barrier 2): poll done, loads done, scalars published
for name, b0, ref in (("barrier 0", 9, 2), ("barrier 2", 12, 6)):
    f = s[:, b0] != 0
    if f.any():
        d = (s[f][:, [b0, b0 + 1, b0 + 2]] - s[f][:, [ref]]) & 0xFFFFFFFF
        print(name, "folders:", int(f.sum()), " poll matched / partials loaded / published after arrival: mean",
              d.mean(axis=0).astype(int), "p90", np.percentile(d, 90, axis=0).astype(int))
