"""Whole-frame pipeline (graph replay, 8 frames) timed for several MI_ISP_POLL_SLEEP values; needs the measure build:
    make -C taichi_image_amd/csrc EXTRA=-DMI_ISP_MEASURE OBJDIR=../../build/csrc_measure OUT=../lib/libmi355_isp_measure.so
    MI_ISP_LIB=taichi_image_amd/lib/libmi355_isp_measure.so python scripts/wf_sweep.py 0 1 2 4 8"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from taichi_image_amd import synthetic
from taichi_image_amd.pipeline import BatchPipeline
H, W = 3072, 4096
dev = torch.device("cuda", 0)
host = [synthetic.synthetic_packed12(i) for i in range(4)]
frames = [torch.from_numpy(host[i % 4]).to(dev) for i in range(8)]
VAR = os.environ.get("SWEEP_VAR", "MI_ISP_POLL_SLEEP")
for val in (sys.argv[1:] or ["1"]):
    os.environ[VAR] = val
    bp = BatchPipeline(8, H, W, dev, use_graph=True, whole_frame=True)      # the graph is captured with this setting
    bp.prepare(frames)
    res = []
    for rep in range(3):
        for _ in range(10): bp(frames)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(150): bp(frames)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 1200 * 1e6)
    print(f"{VAR} {val}: " + " ".join(f"{r:.1f}" for r in res) + " us/frame", flush=True)
    del bp
