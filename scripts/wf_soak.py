"""Soak test of the whole-frame kernel's grid barriers: many launches on one workspace, frames of both kinds of bounds
and two geometries interleaved, every output compared bit for bit with the first output of its frame (the kernel is
deterministic: fixed fold order) and the error word checked.   usage: wf_soak.py [launches]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from taichi_image_amd import synthetic, _native
from taichi_image_amd.pipeline import pipeline12_reinhard, BatchPipeline
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
dev = torch.device("cuda", 0)
H, W = 3072, 4096
def rescale(p, Hh, Ww):
    b = p.reshape(Hh, -1, 3).astype(np.uint32)
    v = np.stack([b[..., 0] | ((b[..., 1] & 0xF) << 8), (b[..., 1] >> 4) | (b[..., 2] << 4)], -1).reshape(Hh, Ww)
    return synthetic.pack12(np.rint(v * 0.7 + 0.1 * 4095).astype(np.uint16))
host = [synthetic.synthetic_packed12(i) for i in range(3)]
frames = [torch.from_numpy(h).to(dev) for h in host] + [torch.from_numpy(rescale(host[0], H, W)).to(dev)]
small_h = synthetic.synthetic_packed12(5, 1536, 2048)
frames.append(torch.from_numpy(small_h).to(dev))
shapes = [(H, W)] * 4 + [(1536, 2048)]
first = [pipeline12_reinhard(f, whole_frame=True).clone() for f in frames]
torch.cuda.synchronize()
outs = [torch.empty_like(o) for o in first]
off = int(_native.lib().mi_isp_workspace_error_offset(H, W))
bad = 0
t0 = time.time()
rng = np.random.default_rng(0)
for it in range(N):
    k = int(rng.integers(0, len(frames)))
    pipeline12_reinhard(frames[k], out=outs[k], whole_frame=True)
    if it % 50 == 49 or it == N - 1:
        torch.cuda.synchronize()
        for j in range(len(frames)):
            if not torch.equal(outs[j], first[j]) and outs[j].abs().sum() != 0:
                bad += 1
                print(f"launch {it}: frame {j} differs from its first output", flush=True)
        for (hh, ww) in set(shapes):
            ws = _native.workspace(hh, ww, dev)
            e = int(ws[off:off + 4].view(torch.int32).item())
            if e:
                bad += 1
                print(f"launch {it}: error word set ({hh}x{ww})", flush=True)
                ws[off:off + 4].zero_()
    if it % 5000 == 4999: print(f"{it + 1} launches, {time.time() - t0:.0f} s, {bad} problems", flush=True)
# graph replays of a batch, too
bp = BatchPipeline(8, H, W, dev, use_graph=True, whole_frame=True)
fr8 = [frames[i % 4] for i in range(8)]
bp.prepare(fr8)
ref8 = [o.clone() for o in bp(fr8)]
torch.cuda.synchronize()
for it in range(N // 8):
    o8 = bp(fr8)
    if it % 100 == 99:
        torch.cuda.synchronize()
        for j in range(8):
            if not torch.equal(o8[j], ref8[j]):
                bad += 1
                print(f"graph replay {it}: frame {j} differs", flush=True)
torch.cuda.synchronize()
print(f"soak done: {N} launches + {N // 8} graph replays of 8, {bad} problems, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
