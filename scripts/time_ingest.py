"""PCIe-inclusive rate of config 2: frames start in host memory, go through ingest.UploadRing (pinned
staging, copy stream) and the fused pipeline; reported next to the HBM-resident rate of bench.py.
   pinned   : the source already sits in the ring's pinned buffers (camera SDK / readinto) - pure H2D
   pageable : the source is an ordinary numpy array, copied into the pinned slot first (host memcpy)"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from taichi_image_amd import ingest, synthetic
from taichi_image_amd.pipeline import pipeline12_reinhard
H, W = 3072, 4096
dev = torch.device("cuda", 0)
src = [synthetic.synthetic_packed12(k) for k in range(2)]
nbytes = src[0].size
n_slots, n = 4, 64
ring = ingest.UploadRing(n_slots, nbytes, dev)
outs = [torch.empty((H, W, 3), dtype=torch.float16, device=dev) for _ in range(n_slots)]
def run(pageable):
    for s in ring.slots:                                # pre-fill for the pinned case
        s.host[:] = src[0].reshape(-1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        slot = ring.acquire()
        if pageable:
            slot.host[:] = src[i & 1].reshape(-1)
        d = slot.commit()
        pipeline12_reinhard(d.view(H, W * 3 // 2), out=outs[i % n_slots], whole_frame=False)
        slot.release()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return n * H * W / 1e6 / dt, dt / n * 1e6, n * nbytes / dt / 1e9
for mode in (False, True):
    run(mode)
    mps, us, gbs = run(mode)
    print(f"{'pageable' if mode else 'pinned  '} source: {mps:9.0f} MP/s  {us:7.1f} us/frame  H2D {gbs:5.1f} GB/s", flush=True)
