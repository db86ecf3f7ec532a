"""Timeline of a batch-persistent whole-frame launch: per frame, the absolute times (100 MHz counter shared by the chip) at
which the waves pass each phase boundary - where the frames of a batch overlap and where they wait.
    make -C taichi_image_amd/csrc EXTRA="-DMI_STREAM_STAMPS -DMI_ISP_MEASURE -DMI_STAMP_REALTIME" OBJDIR=../../build/csrc_stamps OUT=../lib/libmi355_isp_stamps.so
    MI_ISP_LIB=taichi_image_amd/lib/libmi355_isp_stamps.so python scripts/wf_batch_stamps.py [n_frames]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from taichi_image_amd import _native, synthetic
from taichi_image_amd.pipeline import BatchPipeline
H, W = 3072, 4096
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda", 0)
host = [synthetic.synthetic_packed12(k) for k in range(min(n, 4))]
if os.environ.get("NONUNIT"):
    def rescale(p):
        b = p.reshape(H, -1, 3).astype(np.uint32)
        v = np.stack([b[..., 0] | ((b[..., 1] & 0xF) << 8), (b[..., 1] >> 4) | (b[..., 2] << 4)], -1).reshape(H, W)
        return synthetic.pack12(np.rint(v * 0.7 + 0.1 * 4095).astype(np.uint16))
    host = [rescale(h) for h in host]
frames = [torch.from_numpy(host[k % len(host)]).to(dev) for k in range(n)]
bp = BatchPipeline(n, H, W, dev, whole_frame=True)
for _ in range(5): bp(frames)
torch.cuda.synchronize()
ws_bytes = int(_native.lib().mi_isp_workspace_bytes(H, W))
base = 64 + 48 * 4096
nw = 2048
bp(frames)
torch.cuda.synchronize()
raw = bp.ws.cpu().numpy().view(np.uint32).reshape(n, ws_bytes // 4)
S = np.stack([raw[f, base:base + nw * 16].reshape(nw, 16).astype(np.int64) for f in range(n)])     # [frame][wave][slot]
t0 = S[0, :, 0].min()
names = ["entry / frame start", "phase A done", "record 0 posted", "barrier 0 passed", "statistics (phase B)", "phase C done", "record 2 posted",
         "barrier 2 passed", "phase D done"]
print(f"{n} frames in one launch; us since the first wave's entry: p50 / last wave")
for f in range(n):
    ab = (S[f, :, :9] - t0) / 100.0
    print(f"frame {f}: " + "  ".join(f"{names[i].split()[0]}{' ' + names[i].split()[1] if i in (1, 4, 5, 8) else ''} {np.median(ab[:, i]):.1f}/{ab[:, i].max():.1f}" for i in range(9)))
per = [(np.median(S[f, :, 8]) - np.median(S[f - 1, :, 8])) / 100.0 for f in range(1, n)]
print("frame period (median wave, phase D done to phase D done):", np.round(per, 2))
f = n - 1
ab = (S[f, :, :9] - S[f, :, :1]) / 100.0
d = np.diff(ab, axis=1)
print(f"last frame, per wave, us spent between stamps (p50 / p90): " + "  ".join(f"{names[i + 1]}: {np.median(d[:, i]):.2f}/{np.percentile(d[:, i], 90):.2f}" for i in range(8)))
pro = (S[f, :, 15] - S[f, :, 0]) / 100.0
print(f"last frame: first four rows decoded {np.median(pro):.2f} us after the frame's start (p90 {np.percentile(pro, 90):.2f})")
# the barriers of the last frame, absolute: the last record's post, then (role-0 wave of every block) all records seen, partial
# fold left in LDS, and - in the block whose role 0 finished last - the scalars published
for name, b0, post in (("barrier 0", 9, 2), ("barrier 2", 12, 6)):
    last_post = (S[f, :, post].max() - t0) / 100.0
    m = S[f, :, b0] != 0
    seen = (S[f, m, b0] - t0) / 100.0; left = (S[f, m, b0 + 1] - t0) / 100.0
    pub = S[f, :, b0 + 2]; pub = (pub[pub != 0] - t0) / 100.0
    passed = (S[f, :, post + 1] - t0) / 100.0
    print(f"{name}: last record posted {last_post:.2f}; all records seen p50 {np.median(seen):.2f} max {seen.max():.2f}; partial fold left p50 {np.median(left):.2f} "
          f"max {left.max():.2f}; scalars published p50 {np.median(pub):.2f} max {pub.max():.2f} (n={len(pub)}); waves passed p50 {np.median(passed):.2f} max {passed.max():.2f}")
