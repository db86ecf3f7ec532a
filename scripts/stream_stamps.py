"""Phase timeline of one stream-kernel pass from in-kernel s_memtime stamps (lane 0 of every wave).
Needs the instrumented build:
    make -C taichi_image_amd/csrc EXTRA="-DMI_STREAM_STAMPS -DMI_ISP_MEASURE" OBJDIR=../../build/csrc_stamps OUT=../lib/libmi355_isp_stamps.so
    MI_ISP_LIB=taichi_image_amd/lib/libmi355_isp_stamps.so python scripts/stream_stamps.py [pass]
"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from taichi_image_amd import _native, synthetic, types
from taichi_image_amd.pipeline import pipeline12_reinhard
which = int(sys.argv[1]) if len(sys.argv) > 1 else 0
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
                                           types.f16.code, 1.0, 1.0, 0.0, which, ws.data_ptr(), st.cuda_stream))
for _ in range(5): launch()
torch.cuda.synchronize()
cap = 4096
raw = ws.cpu().numpy().view(np.uint32)
base = 64 + 48 * cap
nw = int(os.environ.get("MI_ISP_STREAM_WAVES", "2048"))
s = raw[base:base + nw * 16].reshape(nw, 16).astype(np.int64)
used = [i for i in range(16) if s[:, i].any()]
t0 = s[:, 0].min()
rel = (s - t0) & 0xFFFFFFFF
print("pass", which, "waves", nw, "stamps used:", used, " kernel span (ticks):", rel[:, used].max())
names = {0: "entry", 1: "prologue loads issued", 2: "first 4 rows decoded", 15: "loop done"}
prev = used[0]
print(f"{'phase':28s}   mean    p10    p50    p90   [ticks]")
for i in used[1:]:
    d = rel[:, i] - rel[:, prev]
    print(f"{names.get(i, 'body %d done' % (i - 3)):28s} {d.mean():7.0f} {np.percentile(d,10):6.0f} {np.percentile(d,50):6.0f} {np.percentile(d,90):6.0f}")
    prev = i
life = rel[:, used[-1]] - rel[:, 0]
print(f"{'wave lifetime':28s} {life.mean():7.0f} {np.percentile(life,10):6.0f} {np.percentile(life,50):6.0f} {np.percentile(life,90):6.0f}")
print("wave start times p0/p25/p50/p75/p100 =", [int(np.percentile(rel[:, 0], q)) for q in (0, 25, 50, 75, 100)])
print("wave end times   p0/p25/p50/p75/p100 =", [int(np.percentile(rel[:, used[-1]], q)) for q in (0, 25, 50, 75, 100)])
# edge bands vs interior bands (8 bands per row band)
band = np.arange(nw) % 8
for b in (0, 3, 7):
    print(f"band {b}: mean lifetime {life[band == b].mean():.0f}")
