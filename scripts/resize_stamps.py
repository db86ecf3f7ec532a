"""Phase timeline of the streaming resize kernel from in-kernel s_memtime stamps (lane 0 of every wave).
    make -C taichi_image_amd/csrc EXTRA="-DMI_STREAM_STAMPS -DMI_ISP_MEASURE" OBJDIR=../../build/csrc_stamps OUT=../lib/libmi355_isp_stamps.so
    MI_ISP_LIB=taichi_image_amd/lib/libmi355_isp_stamps.so python scripts/resize_stamps.py
"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import taichi_image_amd as ti
from taichi_image_amd import _native, synthetic
dev = torch.device("cuda", 0)
frame = torch.from_numpy(synthetic.synthetic_packed12(0)).to(dev)
isp = ti.Camera16(ti.BayerPattern.RGGB, resize_width=1920, device=dev)
for _ in range(3): isp.load_packed12(frame)
torch.cuda.synchronize()
L = ctypes.CDLL(_native.LIB_PATH)
nw = 1980
buf = np.zeros((nw, 16), np.uint32)
L.mi_isp_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), nw)      # clears the buffer
isp.load_packed12(frame)
L.mi_isp_debug_read_stamps(buf.ctypes.data_as(ctypes.c_void_p), nw)
s = buf.astype(np.int64)
names = ["entry", "prologue (table, first rows decoded)", "pair 3 starts", "pair 3: two rows demosaiced + in the ring", "pair 3: destination rows emitted", "done"]
rel = (s - s[:, :1]) & 0xFFFFFFFF
ok = s[:, 5] != 0
print("waves with stamps:", int(ok.sum()))
for i in (1, 2, 3, 4, 5):
    d = (rel[ok, i] - rel[ok, i - 1])
    print(f"{names[i]:48s} mean {d.mean():8.0f}  p10 {np.percentile(d,10):7.0f}  p50 {np.percentile(d,50):7.0f}  p90 {np.percentile(d,90):7.0f}")
print("wave lifetime mean", rel[ok, 5].mean())
