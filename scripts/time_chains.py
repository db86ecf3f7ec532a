"""Short timings of the other chains (per 4K frame): multi-pass config 2, the 6-camera ISP step at full resolution and at
resize_width=1920, config 3b.  For A/B runs of library variants: [MI_ISP_LIB=...] python scripts/time_chains.py [label]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import taichi_image_amd as ti
from taichi_image_amd import synthetic, types, interpolate
from taichi_image_amd.pipeline import BatchPipeline
label = sys.argv[1] if len(sys.argv) > 1 else os.path.basename(os.environ.get("MI_ISP_LIB", "default"))
H, W = 3072, 4096
dev = torch.device("cuda", 0)
frames = [torch.from_numpy(synthetic.synthetic_packed12(k % 4)).to(dev) for k in range(8)]
def timed(fn, reps, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
res = []
bp = BatchPipeline(8, H, W, dev, n_streams=2, use_graph=True, whole_frame=False); bp.prepare(frames)
res.append(f"multi-pass {min(timed(lambda: bp(frames), 100) for _ in range(3)) / 8 * 1e6:.2f}")
del bp
for rw in (0, 1920):
    isp = ti.Camera16(ti.BayerPattern.RGGB, moving_alpha=0.1, resize_width=rw, device=dev)
    step = lambda: isp.tonemap_reinhard([isp.load_packed12(f) for f in frames[:6]], gamma=0.6)
    res.append(f"isp6 rw={rw} {min(timed(step, 40) for _ in range(3)) / 6 * 1e6:.2f}")
    res.append(f"load x6 rw={rw} {min(timed(lambda: [isp.load_packed12(f) for f in frames[:6]], 40) for _ in range(3)) / 6 * 1e6:.2f}")
isp = ti.Camera16(ti.BayerPattern.RGGB, device=dev)
step = lambda: [interpolate.resize_bilinear(isp.load_packed12(f), (1920, 1080), scale=(0.3515625, 0.46875), dtype=types.u8) for f in frames[:4]]
res.append(f"config3b u8 {min(timed(step, 40) for _ in range(3)) / 4 * 1e6:.2f}")
print(f"{label}: us per frame: " + "; ".join(res), flush=True)
