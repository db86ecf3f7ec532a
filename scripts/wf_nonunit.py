"""Whole-frame pipeline on frames whose bounds are not (0, 1): graph replay and launch by launch, several timings."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from taichi_image_amd import synthetic, _native
from taichi_image_amd.pipeline import BatchPipeline
H, W = 3072, 4096
dev = torch.device("cuda", 0)
def rescale(p):
    b = p.reshape(H, -1, 3).astype(np.uint32)
    v = np.stack([b[..., 0] | ((b[..., 1] & 0xF) << 8), (b[..., 1] >> 4) | (b[..., 2] << 4)], -1).reshape(H, W)
    return synthetic.pack12(np.rint(v * 0.7 + 0.1 * 4095).astype(np.uint16))
host = [synthetic.synthetic_packed12(i) for i in range(4)]
unit = [torch.from_numpy(host[i % 4]).to(dev) for i in range(8)]
nonunit = [torch.from_numpy(rescale(host[i % 4])).to(dev) for i in range(8)]
def t(bp, fr, eager, steps=100):
    for _ in range(5): bp(fr, eager=eager)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): bp(fr, eager=eager)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (steps * 8) * 1e6
off = int(_native.lib().mi_isp_workspace_error_offset(H, W))
for name, fr in (("unit", unit), ("non-unit", nonunit), ("unit again", unit), ("non-unit again", nonunit)):
    bp = BatchPipeline(8, H, W, dev, use_graph=True, whole_frame=True)
    bp.prepare(fr)
    g = [t(bp, fr, False) for _ in range(4)]
    e = [t(bp, fr, True) for _ in range(2)]
    wsb = bp.ws.numel() // 8
    err = [int(bp.ws[k * wsb + off: k * wsb + off + 4].view(torch.int32).item()) for k in range(8)]
    print(f"{name}: graph " + " ".join(f"{x:.1f}" for x in g) + "  eager " + " ".join(f"{x:.1f}" for x in e) + f"  error flags {err}", flush=True)
    del bp
