"""Soak test of the batch-persistent whole-frame launch: 64-frame launches (frames of both kinds of bounds interleaved)
on one workspace, every output of every launch compared bit for bit with the first launch's (the kernel is
deterministic: fixed fold order), the fault words and the mailbox checked.   usage: wf_soak_batch.py [launches] [frames]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from taichi_image_amd import synthetic, _native
from taichi_image_amd.pipeline import BatchPipeline
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda", 0)
H, W = 3072, 4096
def rescale(p):
    b = p.reshape(H, -1, 3).astype(np.uint32)
    v = np.stack([b[..., 0] | ((b[..., 1] & 0xF) << 8), (b[..., 1] >> 4) | (b[..., 2] << 4)], -1).reshape(H, W)
    return synthetic.pack12(np.rint(v * 0.7 + 0.1 * 4095).astype(np.uint16))
host = [synthetic.synthetic_packed12(i) for i in range(3)]
host.append(rescale(host[0]))
dist = [torch.from_numpy(h).to(dev) for h in host]
frames = [dist[(3 * k + k // 5) % 4] for k in range(n)]
bp = BatchPipeline(n, H, W, dev, whole_frame=True, on_timeout="raise")
first = [o.clone() for o in bp(frames)]
torch.cuda.synchronize()
assert bp.check(frames) == []
bad = 0
t0 = time.time()
for it in range(N):
    outs = bp(frames)
    if it % 20 == 19 or it == N - 1:
        torch.cuda.synchronize()
        for j in range(n):
            if not torch.equal(outs[j], first[j]):
                bad += 1
                print(f"launch {it}: frame {j} differs from the first launch's", flush=True)
        failed = bp.check(frames)
        if failed:
            bad += 1
            print(f"launch {it}: barrier timeouts in frames {failed}", flush=True)
    if it % 500 == 499:
        print(f"{it + 1} launches, {(it + 1) * n} frames, {time.time() - t0:.0f} s, bad {bad}", flush=True)
torch.cuda.synchronize()
dt = time.time() - t0
print(f"{N} launches of {n} frames = {N * n} frames in {dt:.1f} s ({dt / (N * n) * 1e6:.1f} us per frame incl. checks): "
      f"{bad} bad, mailbox {int(_native.lib().mi_isp_whole_frame_faults(0))}")
sys.exit(1 if bad else 0)
