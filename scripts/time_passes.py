"""Isolated time of each data pass of the cached config-2 pipeline (events on the stream)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from taichi_image_amd import _native, synthetic, types
from taichi_image_amd.pipeline import pipeline12_reinhard
H, W = 3072, 4096
dev = torch.device("cuda", 0)
frame = torch.from_numpy(synthetic.synthetic_packed12(0)).to(dev)
out = torch.empty((H, W, 3), dtype=torch.float16, device=dev)
pipeline12_reinhard(frame, out=out, whole_frame=False)
ws = _native.workspace(H, W, dev)
L = _native.lib()
st = torch.cuda.current_stream(dev)
def t(code, reps=50):
    def launch():
        _native.check(L.mi_isp_pipeline12_pass(frame.data_ptr(), out.data_ptr(), H, W, 0, 0, None, types.f16.code,
                                               types.f16.code, 1.0, 1.0, 0.0, code, ws.data_ptr(), st.cuda_stream))
    for _ in range(10): launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps): launch()
    e1.record(st); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
print("PULL_DEBUG =", os.environ.get("MI_ISP_PULL_DEBUG"), " ".join(f"pass{p}: {t(p):.1f} us" for p in range(4)), flush=True)
def full(reps=50):
    for _ in range(5): pipeline12_reinhard(frame, out=out, whole_frame=False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps): pipeline12_reinhard(frame, out=out, whole_frame=False)
    e1.record(st); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
print(f"whole frame, one stream: {full():.1f} us", flush=True) if not os.environ.get("MI_ISP_PULL_DEBUG") else None
