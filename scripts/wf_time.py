"""Per-frame time of the whole-frame kernel as a replayed batch (what bench.py's headline runs), unit and non-unit frames.
    [MI_ISP_LIB=...] python scripts/wf_time.py [n_frames] [label]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from taichi_image_amd import _native, synthetic, types
from taichi_image_amd.pipeline import BatchPipeline
H, W = 3072, 4096
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
label = sys.argv[2] if len(sys.argv) > 2 else os.path.basename(os.environ.get("MI_ISP_LIB", "default"))
dev = torch.device("cuda", 0)
host = [synthetic.synthetic_packed12(k) for k in range(min(n, 4))]
def rescale(p):
    b = p.reshape(H, -1, 3).astype(np.uint32)
    v = np.stack([b[..., 0] | ((b[..., 1] & 0xF) << 8), (b[..., 1] >> 4) | (b[..., 2] << 4)], -1).reshape(H, W)
    return synthetic.pack12(np.rint(v * 0.7 + 0.1 * 4095).astype(np.uint16))
sets = {"unit": [torch.from_numpy(host[k % len(host)]).to(dev) for k in range(n)],
        "non-unit": [torch.from_numpy(rescale(host[k % 2])).to(dev) for k in range(n)]}
def timed(fn, reps, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
res = []
for name, fr in sets.items():
    for dt in ((types.f16, types.u8) if os.environ.get("U8") else (types.f16,)):
        bg = BatchPipeline(n, H, W, dev, whole_frame=True, use_graph=True, dtype=dt)
        bg.prepare(fr)
        ts = [timed(lambda: bg(fr), 300) / n * 1e6 for _ in range(3)]
        res.append(f"{name}{'/u8' if dt is types.u8 else ''} {min(ts):.2f} ({', '.join(f'{t:.2f}' for t in ts)})")
        del bg
print(f"{label}: us per frame, batch of {n}, graph replay: " + "; ".join(res) + f"; faults {int(_native.lib().mi_isp_whole_frame_faults(0))}", flush=True)
