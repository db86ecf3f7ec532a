"""ISP.tonemap_reinhard on six full-resolution images, gamma 0.6 against gamma 1 (is pass 2 bound by its pow?)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import taichi_image_amd as ti
from taichi_image_amd import synthetic
dev = torch.device("cuda", 0)
frames = [torch.from_numpy(synthetic.synthetic_packed12(i)).to(dev) for i in range(6)]
isp = ti.Camera16(ti.BayerPattern.RGGB, moving_alpha=0.1, device=dev)
st = torch.cuda.current_stream(dev)
for gamma in (0.6, 1.0, 0.6):
    imgs = [isp.load_packed12(f) for f in frames]
    keep = [im.clone() for im in imgs]
    for _ in range(3): isp.tonemap_reinhard([k.clone() for k in keep], gamma=gamma)
    work = [[k.clone() for k in keep] for _ in range(10)]
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for w in work: isp.tonemap_reinhard(w, gamma=gamma)
    e1.record(st); e1.synchronize()
    print(f"gamma {gamma}: tonemap_reinhard of 6 full-resolution images {e0.elapsed_time(e1) * 1e3 / 10:.1f} us per call", flush=True)
