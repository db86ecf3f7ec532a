"""Where the one-launch metering kernel spends its time (build with EXTRA=-DMI_METER_STAMPS): 100 MHz stamps of thread 0 of
every block.   MI_ISP_LIB=.../libv_mstamps.so python scripts/meter_stamps.py"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import taichi_image_amd as ti
from taichi_image_amd import _native, synthetic
dev = torch.device("cuda", 0)
H, W = 3072, 4096
frames = [torch.from_numpy(synthetic.synthetic_packed12(k)).to(dev) for k in range(6)]
cam = ti.Camera16(ti.BayerPattern.RGGB, moving_alpha=0.3, device=dev)
for it in range(3):
    imgs = [cam.load_packed12(f) for f in frames]
    cam.update_metering(imgs)
torch.cuda.synchronize()
cap = 4096
for hh, ww in ((H // 8, W // 8), (H, W)):
    ws = _native.workspace(hh, ww, dev)
    q = ws.view(torch.int64).cpu().numpy()
    base = (64 + cap + 3 * 1024 * 4) * 4 // 8
    S = q[base:base + 256 * 8].reshape(256, 8)
    S = S[S[:, 0] != 0]
    if len(S) == 0: continue
    t0 = S[:, 0].min()
    print((hh, ww), len(S), "blocks; us since the first block's start (median / max over blocks):")
    for i, name in enumerate(("start", "pass 1 done + posted", "bounds known", "pass 2 done + posted")):
        print(f"   {name:24s} {np.median(S[:, i] - t0) / 100:.2f} / {(S[:, i] - t0).max() / 100:.2f}")
    print(f"   pass 2 loop done          {np.median(S[:, 6] - t0) / 100:.2f} / {(S[:, 6] - t0).max() / 100:.2f};  waves reduced, block met {np.median(S[:, 7] - t0) / 100:.2f} / {(S[:, 7] - t0).max() / 100:.2f}")
    b0 = S[0]
    print("   block 0: records of pass 2 seen", (b0[4] - t0) / 100, "state written", (b0[5] - t0) / 100)
