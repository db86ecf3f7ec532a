"""Who is late?  From the stamps of a batch launch (scripts/wf_batch_stamps.py's build): the lateness of every wave at the
end of phase D and of phase A (relative to the median wave of the frame), averaged over frames, grouped by XCD
(block % 8), by band column (bx) and by band row (by) - is the spread of the arrivals at barrier 0 a property of places?
    MI_ISP_LIB=taichi_image_amd/lib/libmi355_isp_stamps.so python scripts/wf_skew.py [n_frames]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from taichi_image_amd import _native, synthetic
from taichi_image_amd.pipeline import BatchPipeline
H, W = 3072, 4096
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
host = [synthetic.synthetic_packed12(k) for k in range(4)]
if os.environ.get("FLIP"):          # the same scenes upside down (row pairs kept: the CFA pattern stays RGGB): place or data?
    host = [np.ascontiguousarray(h.reshape(H // 2, 2, -1)[::-1].reshape(H, -1)) for h in host]
if os.environ.get("FLAT"):          # a featureless frame: every pixel the same code
    host = [synthetic.pack12(np.full((H, W), 1000 + 100 * k, np.uint16)) for k in range(4)]
frames = [torch.from_numpy(host[k % 4]).to(dev) for k in range(n)]
bp = BatchPipeline(n, H, W, dev, whole_frame=True)
for _ in range(3): bp(frames)
torch.cuda.synchronize()
ws_bytes = int(_native.lib().mi_isp_workspace_bytes(H, W))
base, nw = 64 + 48 * 4096, 2048
bp(frames); torch.cuda.synchronize()
raw = bp.ws.cpu().numpy().view(np.uint32).reshape(n, ws_bytes // 4)
S = np.stack([raw[f, base:base + nw * 16].reshape(nw, 16).astype(np.int64) for f in range(n)]) / 100.0      # us
g = np.arange(nw); block = g // 4; bx = g % 8; by = g // 8; xcd = block % 8
# (the stamps are indexed by g = the wave's place in the image; with -DMI_MEGA_TEST_PERMUTE the dispatch index of a block is
# bid = 2 * (blk % 256) + blk // 256)
bid = 2 * (block % 256) + block // 256 if os.environ.get("PERMUTED") else block
for name, slot in (("end of phase D", 8), ("end of phase A", 1), ("record 0 posted", 2)):
    late = np.stack([S[f, :, slot] - np.median(S[f, :, slot]) for f in range(1, n)]).mean(axis=0)      # per wave, mean over frames
    rep = np.stack([S[f, :, slot] - np.median(S[f, :, slot]) for f in range(1, n)])
    print(f"{name}: lateness vs the frame's median wave [us]: p10 {np.percentile(late,10):+.2f} p50 {np.percentile(late,50):+.2f} p90 {np.percentile(late,90):+.2f} max {late.max():+.2f}; "
          f"single frames: p90 {np.percentile(rep,90):+.2f} max {rep.max():+.2f}; correlation of a wave's lateness between consecutive frames {np.corrcoef(rep[:-1].ravel(), rep[1:].ravel())[0,1]:+.2f}")
    print("   by XCD (block % 8):", " ".join(f"{late[xcd == x].mean():+.2f}" for x in range(8)))
    print("   by band column bx: ", " ".join(f"{late[bx == x].mean():+.2f}" for x in range(8)))
    q = [late[(by >= 32 * k) & (by < 32 * k + 32)].mean() for k in range(8)]
    print("   by band row (eighths of the image, top to bottom):", " ".join(f"{x:+.2f}" for x in q), f"  first band {late[by == 0].mean():+.2f} last band {late[by == 255].mean():+.2f}")
    print(f"   by dispatch index: blocks 0..255 {late[bid < 256].mean():+.2f}, blocks 256..511 {late[bid >= 256].mean():+.2f}; by image half: top {late[by < 128].mean():+.2f}, bottom {late[by >= 128].mean():+.2f}")
    worst = np.argsort(late)[-8:]
    print("   latest waves (wave, block, bx, by, late):", [(int(w), int(block[w]), int(bx[w]), int(by[w]), round(float(late[w]), 2)) for w in worst])
