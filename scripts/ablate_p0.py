"""Ablation of the cached pipeline's pass 0 (tile_kernel<EPI_STORE_MINMAX>): debug_skip bits."""
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
def t(skip, reps=50):
    code = (skip | 64) << 4
    def launch():
        _native.check(L.mi_isp_pipeline12_pass(frame.data_ptr(), out.data_ptr(), H, W, 0, 0, None, types.f16.code,
                                               types.f16.code, 1.0, 1.0, 0.0, code, ws.data_ptr(), st.cuda_stream))
    for _ in range(10): launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(reps): launch()
    e1.record(st); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
names = {0: "full", 1: "no fill", 2: "no compute", 4: "no reduce", 8: "no accumulate", 16: "no fixup", 32: "no store",
         1 | 32: "no fill, no store", 2 | 32: "no compute, no store", 1 | 2: "no fill, no compute (stores only)",
         1 | 2 | 4: "stores only, no reduce", 8 | 16: "no accumulate/fixup", 4 | 32: "no reduce, no store", 1|2|4|32: "nothing"}
for k, v in names.items():
    print(f"{v:40s} {t(k):7.1f} us", flush=True)
