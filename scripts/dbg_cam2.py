import os, sys, ctypes
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import taichi_image_amd as ti
from taichi_image_amd import _native
from taichi_image_amd.synthetic import synthetic_scene, mosaic_rggb, pack12
dev = torch.device("cuda", 0)
L = _native.lib()
GAINS = [1.0, 0.55, 0.8, 0.3, 0.95, 0.7]; OFFSETS = [0.0, 0.02, 0.1, 0.0, 0.04, 0.15]
scenes = [mosaic_rggb(synthetic_scene(k)) for k in range(3)]
def packed_from(cfa, gain=1.0, offset=0.0):
    v12 = np.rint(np.clip(cfa.astype(np.float64) * gain + offset, 0, 1) * 4095).astype(np.uint16)
    return pack12(v12)
H, W, n = 3072, 4096, 3
for variant in ("images_in_group1", "images_always", "never"):
    a = ti.Camera16(ti.BayerPattern.RGGB, moving_alpha=0.25, device=dev)
    b = ti.Camera16(ti.BayerPattern.RGGB, moving_alpha=0.25, device=dev)
    for group in range(3):
        packs = [torch.from_numpy(packed_from(scenes[k], GAINS[(k + group) % 6], OFFSETS[(k + group) % 6])).to(dev) for k in range(n)]
        keep = variant == "images_always" or (variant == "images_in_group1" and group == 1)
        got = a.process_packed12(packs, gamma=0.6, intensity=1.2, light_adapt=0.8, keep_images=keep)
        outs, imgs = got if keep else (got, None)
        wi = [b.load_packed12(p) for p in packs]
        want = b.tonemap_reinhard(wi, gamma=0.6, intensity=1.2, light_adapt=0.8)
        torch.cuda.synchronize()
        print(variant, "group", group, "metrics", torch.equal(a.metrics, b.metrics), "faults", L.mi_isp_camera_group_faults(0))
        for k in range(n):
            d = (outs[k].int() - want[k].int())
            nz = torch.nonzero(d)
            print("   cam", k, "u8 diffs", int((d != 0).sum()), "max", int(d.abs().max()), "first", nz[:3].tolist() if len(nz) else "",
                  "p diffs", int((imgs[k].view(torch.int16) != wi[k].view(torch.int16)).sum()) if keep else "-")
