"""process_packed12 (one persistent launch per camera group) against load_packed12_batch + tonemap_reinhard, bit for bit."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import taichi_image_amd as ti
from taichi_image_amd import _native, synthetic
dev = torch.device("cuda", 0)
L = _native.lib()
def frames_for(h, w, n, seed):
    rng = np.random.default_rng(seed)
    return [torch.from_numpy(rng.integers(0, 256, (h, w * 3 // 2), dtype=np.uint8)).to(dev) for _ in range(n)]
bad = 0
for (h, w, n, pat, cc, gamma, ca) in [(48, 64, 3, ti.BayerPattern.RGGB, False, 1.0, 0.0), (36, 520, 2, ti.BayerPattern.GRBG, True, 0.6, 0.0),
                                      (100, 1032, 4, ti.BayerPattern.BGGR, True, 0.6, 0.3), (3072, 4096, 6, ti.BayerPattern.RGGB, True, 0.6, 0.0)]:
    a = ti.Camera16(pat, correct_colors=cc, device=dev)
    b = ti.Camera16(pat, correct_colors=cc, device=dev)
    print("fits", h, w, L.mi_isp_camera_group_fits(h, w, pat.value, 2, 8))
    for step in range(3):
        fr = frames_for(h, w, n, 100 * step + h)
        outs, imgs = a.process_packed12(fr, gamma=gamma, color_adapt=ca, keep_images=True)
        outs2 = a.__class__.process_packed12  # noqa
        ref_imgs = b.load_packed12_batch(fr)
        ref_outs = b.tonemap_reinhard(ref_imgs, gamma=gamma, color_adapt=ca)
        torch.cuda.synchronize()
        ok_o = all(torch.equal(x, y) for x, y in zip(outs, ref_outs))
        ok_i = all(torch.equal(x.view(torch.int16), y.view(torch.int16)) for x, y in zip(imgs, ref_imgs))
        ok_m = torch.equal(a.metrics.view(torch.int32), b.metrics.view(torch.int32))
        nd = [int((x.int() - y.int()).abs().max()) for x, y in zip(outs, ref_outs)]
        print(h, w, pat.name, "step", step, "outs", ok_o, nd, "images", ok_i, "metrics", ok_m, a.metrics.cpu().numpy()[:4], "faults", L.mi_isp_camera_group_faults(0))
        bad += not (ok_o and ok_i and ok_m)
# timing at 4K: 6 cameras
a = ti.Camera16(ti.BayerPattern.RGGB, device=dev)
b = ti.Camera16(ti.BayerPattern.RGGB, device=dev)
fr = [torch.from_numpy(synthetic.synthetic_packed12(i)).to(dev) for i in range(6)]
for keep in (False, True):
    for _ in range(5): a.process_packed12(fr, gamma=0.6, keep_images=keep)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): a.process_packed12(fr, gamma=0.6, keep_images=keep)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    print(f"process_packed12 keep_images={keep}: {dt * 1e3:.3f} ms per 6-camera step = {dt / 6 * 1e6:.1f} us per frame")
for _ in range(5): b.tonemap_reinhard(b.load_packed12_batch(fr), gamma=0.6)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): b.tonemap_reinhard(b.load_packed12_batch(fr), gamma=0.6)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
print(f"load_packed12_batch + tonemap_reinhard: {dt * 1e3:.3f} ms per step = {dt / 6 * 1e6:.1f} us per frame")
print("BAD" if bad else "ALL EQUAL")
