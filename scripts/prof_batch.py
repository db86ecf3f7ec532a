"""Profiling driver: a few launches of the headline configuration - 64 unit-bounds 4K frames through ONE launch of the
whole-frame kernel (what bench.py times) - and a few of the same with non-unit bounds, for rocprofv3 --pmc passes.
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d OUT -- python3 scripts/prof_batch.py [n_frames] [launches]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from taichi_image_amd import synthetic
from taichi_image_amd.pipeline import BatchPipeline
H, W = 3072, 4096
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
host = [synthetic.synthetic_packed12(k) for k in range(4)]
frames = [torch.from_numpy(host[k % 4]).to(dev) for k in range(n)]
bp = BatchPipeline(n, H, W, dev, whole_frame=True)
for _ in range(reps):
    bp(frames)
torch.cuda.synchronize()
assert bp.check(frames) == []
print("done")
