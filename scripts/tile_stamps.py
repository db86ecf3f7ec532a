"""Phase timeline of the packed-RAW tile kernel (pass 0) from in-kernel s_memtime stamps.
Needs the instrumented build:
    make -C taichi_image_amd/csrc EXTRA=-DMI_TILE_STAMPS OBJDIR=../../build/csrc_stamps OUT=../lib/libmi355_isp_stamps.so
    MI_ISP_LIB=taichi_image_amd/lib/libmi355_isp_stamps.so python scripts/tile_stamps.py
"""
import os, sys
import numpy as np
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
def launch():
    _native.check(L.mi_isp_pipeline12_pass(frame.data_ptr(), out.data_ptr(), H, W, 0, 0, None, types.f16.code,
                                           types.f16.code, 1.0, 1.0, 0.0, 0, ws.data_ptr(), st.cuda_stream))
for _ in range(5): launch()
torch.cuda.synchronize()
cap = max(4096, (W // 128) * (H // 32))
raw = ws.cpu().numpy().view(np.uint32)
base = 64 + 2 * cap
nt = (W // 128) * (H // 32)
s = raw[base:base + nt * 8].reshape(nt, 8).astype(np.int64)
t0 = s[:, 0].min()
rel = (s - t0) & 0xFFFFFFFF
names = ["start", "fill issued+unpacked", "fill barrier", "window+barrier", "row0 computed", "row1 computed", "strips done", "reduced"]
print("kernel span (cycles):", rel.max(), " tiles:", nt)
print("phase (wave 0 of each block)        mean   p10   p50   p90  [cycles]")
for i in range(1, 8):
    d = rel[:, i] - rel[:, i - 1]
    print(f"{names[i]:32s} {d.mean():8.0f} {np.percentile(d,10):6.0f} {np.percentile(d,50):6.0f} {np.percentile(d,90):6.0f}")
life = rel[:, 7] - rel[:, 0]
print(f"{'block lifetime':32s} {life.mean():8.0f} {np.percentile(life,10):6.0f} {np.percentile(life,50):6.0f} {np.percentile(life,90):6.0f}")
starts = np.sort(rel[:, 0])
print("block start times: p25/p50/p75/p100 =", [int(np.percentile(starts, q)) for q in (25, 50, 75, 100)])
print("concurrency estimate: sum(lifetime)/span =", life.sum() / rel.max(), "blocks in flight (of", 256 * 4, "slots at 4/CU)")
