"""process_packed12 on 6 full-resolution cameras: HIP-event time per step for gamma / keep_images variants, and the
two-call reference sequence beside it."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import taichi_image_amd as ti
from taichi_image_amd import synthetic
dev = torch.device("cuda", 0)
fr = [torch.from_numpy(synthetic.synthetic_packed12(i)).to(dev) for i in range(6)]
def timed(fn, n=60, warm=8):
    for _ in range(warm): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
a = ti.Camera16(ti.BayerPattern.RGGB, device=dev)
b = ti.Camera16(ti.BayerPattern.RGGB, device=dev)
for gamma in (0.6, 1.0):
    for keep in (False, True):
        us = timed(lambda: a.process_packed12(fr, gamma=gamma, keep_images=keep))
        print(f"process_packed12 gamma={gamma} keep_images={keep}: {us:.1f} us per step = {us / 6:.1f} us per frame")
    us = timed(lambda: b.tonemap_reinhard(b.load_packed12_batch(fr), gamma=gamma))
    print(f"load_packed12_batch + tonemap_reinhard gamma={gamma}: {us:.1f} us per step = {us / 6:.1f} us per frame")
