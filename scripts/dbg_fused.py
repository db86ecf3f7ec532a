import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import taichi_image_amd as ti
from tests.util import natural_packed12
from taichi_image_amd import _native
dev = torch.device("cuda", 0)
for (H, W, n) in ((48, 64, 1), (96, 128, 2), (768, 1024, 3), (1440, 1920 * 1, 6)):
    Hs, Ws = (H, W)
    frames = [torch.from_numpy(natural_packed12(np.random.default_rng(700 + k), Hs, Ws, dark=0.03 * k)).to(dev) for k in range(n)]
    res = {}
    for mode in ("1", "2"):
        if mode == "2": os.environ["MI_ISP_REINHARD_LAUNCHES"] = "2"
        else: os.environ["MI_ISP_REINHARD_LAUNCHES"] = "1"
        isp = ti.Camera16(ti.BayerPattern.RGGB, moving_alpha=0.4, device=dev)
        imgs = [isp.load_packed12(f) for f in frames]
        outs = isp.tonemap_reinhard(imgs, gamma=0.6)
        torch.cuda.synchronize()
        ws = _native.workspace(Hs, Ws, dev)
        fp = ws[:256].view(torch.float32).cpu().numpy()
        res[mode] = (imgs, outs, fp.copy())
    for k in range(n):
        a, b = res["1"][1][k], res["2"][1][k]
        pa, pb = res["1"][0][k], res["2"][0][k]
        print(H, W, n, "img", k, "u8 equal", torch.equal(a, b), "mean", a.float().mean().item(), b.float().mean().item(),
              "p equal", torch.equal(pa, pb), "pmax", pa.float().max().item(), pb.float().max().item())
    print("  fp[18..26] fused", res["1"][2][18:26], "two-pass", res["2"][2][18:20], "faults", _native.lib().mi_isp_reinhard_faults(1))
