"""Isolated time of each data pass of the config-2 chain and of load_packed12 (events on the stream)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import taichi_image_amd as ti
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
def timeit(fn, reps=50):
    for _ in range(10): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps): fn()
    e1.record(st); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
def one_pass(code):
    return lambda: _native.check(L.mi_isp_pipeline12_pass(frame.data_ptr(), out.data_ptr(), H, W, 0, 0, None, types.f16.code,
                                                          types.f16.code, 1.0, 1.0, 0.0, code, ws.data_ptr(), st.cuda_stream))
isp = ti.Camera16(ti.BayerPattern.RGGB, device=dev)
tag = os.environ.get("MI_ISP_STREAM_WAVES", "default")
print(f"waves={tag}", " ".join(f"pass{p}: {timeit(one_pass(p)):.1f}" for p in range(4)),
      f"frame: {timeit(lambda: pipeline12_reinhard(frame, out=out, whole_frame=False)):.1f}",
      f"load_packed12: {timeit(lambda: isp.load_packed12(frame)):.1f} us", flush=True)
