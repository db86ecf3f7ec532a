"""Soak of the camera-group kernel: the same sequence of 6-camera steps (rolling metering, frames rotating) twice through
ISP.process_packed12 and once through the two calls; every step's outputs are folded into a checksum on the device.
Equal checksums = no frame was ever mapped with a torn record, a stale max_out or another camera's rows."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import taichi_image_amd as ti
from taichi_image_amd import _native, synthetic
dev = torch.device("cuda", 0)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
pool = [torch.from_numpy(synthetic.synthetic_packed12(i)).to(dev) for i in range(9)]
w = torch.arange(1, 4096 * 3 + 1, device=dev, dtype=torch.int64).view(1, 4096, 3)
def run(kind, n):
    isp = ti.Camera16(ti.BayerPattern.RGGB, moving_alpha=0.1, device=dev)
    acc = torch.zeros((), dtype=torch.int64, device=dev)
    t0 = time.perf_counter()
    for s in range(n):
        fr = [pool[(s + k) % len(pool)] for k in range(6)]
        outs = isp.process_packed12(fr, gamma=0.6) if kind == "one" else isp.tonemap_reinhard(isp.load_packed12_batch(fr), gamma=0.6)
        for k, o in enumerate(outs):
            acc = acc * 31 + (o.to(torch.int64) * w).sum() * (k + 1)
    torch.cuda.synchronize()
    return int(acc.item()), time.perf_counter() - t0, isp.metrics.cpu().numpy()
a, ta, ma = run("one", steps)
b, tb, mb = run("one", steps)
c, tc, mc = run("two", min(steps, 60))
a60, _, _ = run("one", min(steps, 60))
print(f"{steps} steps x 6 cameras: checksums {a} / {b} ({'equal' if a == b else 'DIFFER'}), {ta:.1f} s + {tb:.1f} s; "
      f"first {min(steps, 60)} steps against the two calls: {'equal' if a60 == c else 'DIFFER'}; faults {_native.lib().mi_isp_camera_group_faults(0)}")
print("metrics equal:", bool((ma == mb).all()))
