"""Can events recorded inside a captured HIP graph be timed after a replay?"""
import os, sys, ctypes
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from taichi_image_amd import synthetic, _native
from taichi_image_amd.pipeline import BatchPipeline
H, W = 3072, 4096
dev = torch.device("cuda", 0)
L = _native.lib()
frames = [torch.from_numpy(synthetic.synthetic_packed12(i % 2)).to(dev) for i in range(8)]
bp = BatchPipeline(8, H, W, dev, n_streams=2, whole_frame=False)
bp(frames); torch.cuda.synchronize()
_native.check(L.mi_isp_profile_enable(64, 3))
g = torch.cuda.CUDAGraph()
cap = torch.cuda.Stream(device=dev)
with torch.cuda.stream(cap):
    with torch.cuda.graph(g, stream=cap):
        bp._issue(frames)
torch.cuda.synchronize()
for rep in range(3):
    g.replay(); g.replay()
    torch.cuda.synchronize()
    us, n = (ctypes.c_float * 4)(), ctypes.c_int(0)
    rc = L.mi_isp_profile_collect(us, ctypes.byref(n))
    print("rc", rc, "frames", n.value, [round(float(x), 2) for x in us], flush=True)
    if rc: print(L.mi_isp_last_error().decode()); break
    # collect() resets `used`; re-arm the count without destroying the events
