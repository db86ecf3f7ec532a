"""Run-to-run variance of the whole-frame kernel inside a BatchPipeline (graph replay)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from taichi_image_amd import synthetic
from taichi_image_amd.pipeline import BatchPipeline
H, W = 3072, 4096
dev = torch.device("cuda", 0)
host = [synthetic.synthetic_packed12(i) for i in range(4)]
frames = [torch.from_numpy(host[i % 4]).to(dev) for i in range(8)]
def run(tag, steps=150, **kw):
    bp = BatchPipeline(8, H, W, dev, use_graph=True, **kw)
    bp.prepare(frames)
    for _ in range(10): bp(frames)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): bp(frames)
    torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / (steps * 8) * 1e6
    print(f"{tag}: {us:.1f} us per frame", flush=True)
    return bp
for i in range(3): run(f"whole-frame #{i}", whole_frame=True)
run("multi-pass", n_streams=2)
for i in range(3): run(f"whole-frame after multi-pass #{i}", whole_frame=True)
keep = [run(f"whole-frame keeping the pipelines alive #{i}", whole_frame=True) for i in range(3)]
