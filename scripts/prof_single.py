"""Profiling driver: N back-to-back single-frame runs of the fused config-2 pipeline (one stream),
so that rocprofv3 sees each data pass in isolation.  Usage (on the GPU box):
    rocprofv3 --kernel-trace --stats --output-format csv -d OUT -- python3 scripts/prof_single.py
    rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU ... --output-format csv -d OUT -- python3 scripts/prof_single.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from taichi_image_amd import synthetic  # noqa: E402
from taichi_image_amd.pipeline import pipeline12_reinhard  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
# cheap stand-in for the synthetic scene (same statistics are irrelevant for counters)
frame = torch.from_numpy(synthetic.synthetic_packed12(0)).to(dev)
out = torch.empty((3072, 4096, 3), dtype=torch.float16, device=dev)
for _ in range(n):
    pipeline12_reinhard(frame, out=out, whole_frame=False)
for _ in range(n):                                   # the single-launch kernel of the same chain
    pipeline12_reinhard(frame, out=out, whole_frame=True)
torch.cuda.synchronize()
print("done")
