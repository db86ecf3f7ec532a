"""Profiling driver: one data pass of the cached config-2 pipeline, repeated (for rocprofv3 --pmc).
    rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU ... --output-format csv -d OUT -- python3 scripts/prof_pass.py 0 5
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from taichi_image_amd import _native, synthetic, types  # noqa: E402
from taichi_image_amd.pipeline import pipeline12_reinhard  # noqa: E402

which = int(sys.argv[1]) if len(sys.argv) > 1 else 0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
H, W = 3072, 4096
dev = torch.device("cuda", 0)
frame = torch.from_numpy(synthetic.synthetic_packed12(0)).to(dev)
out = torch.empty((H, W, 3), dtype=torch.float16, device=dev)
pipeline12_reinhard(frame, out=out, whole_frame=False)
ws = _native.workspace(H, W, dev)
st = torch.cuda.current_stream(dev)
for _ in range(reps):
    _native.check(_native.lib().mi_isp_pipeline12_pass(frame.data_ptr(), out.data_ptr(), H, W, 0, 0, None, types.f16.code,
                                                       types.f16.code, 1.0, 1.0, 0.0, which, ws.data_ptr(), st.cuda_stream))
torch.cuda.synchronize()
print("done")
