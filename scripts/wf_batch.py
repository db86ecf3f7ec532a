"""The whole-frame kernel as a batch-persistent launch: parity of every frame of a batch with the single-frame launch and
the C oracle, and the per-frame time launch by launch / as a batch / as a replayed graph.
    python scripts/wf_batch.py [n_frames]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from taichi_image_amd import _native, synthetic, types
from taichi_image_amd.pipeline import pipeline12_reinhard, BatchPipeline
H, W = 3072, 4096
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
host = [synthetic.synthetic_packed12(k) for k in range(min(n, 4))]
def rescale(p):
    b = p.reshape(H, -1, 3).astype(np.uint32)
    v = np.stack([b[..., 0] | ((b[..., 1] & 0xF) << 8), (b[..., 1] >> 4) | (b[..., 2] << 4)], -1).reshape(H, W)
    return synthetic.pack12(np.rint(v * 0.7 + 0.1 * 4095).astype(np.uint16))
host_nu = [rescale(h) for h in host[:2]]
frames = [torch.from_numpy(host[k % len(host)]).to(dev) for k in range(n)]
frames_nu = [torch.from_numpy(host_nu[k % len(host_nu)]).to(dev) for k in range(n)]
mixed = [frames_nu[k] if k % 3 == 1 else frames[k] for k in range(n)]

def timed(fn, reps, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

def faults():
    return int(_native.lib().mi_isp_whole_frame_faults(0))

for name, fr in (("unit", frames), ("non-unit", frames_nu), ("mixed", mixed)):
    singles = [pipeline12_reinhard(f, whole_frame=True).clone() for f in fr[:4]]
    multi = [pipeline12_reinhard(f).clone() for f in fr[:2]]
    torch.cuda.synchronize()
    bp = BatchPipeline(n, H, W, dev, whole_frame=True, use_graph=False)
    outs = bp(fr)
    torch.cuda.synchronize()
    same = [bool(torch.equal(outs[k], singles[k])) for k in range(min(n, 4))]
    d = [(outs[k].float() - multi[k].float()).abs().max().item() for k in range(min(n, 2))]
    print(f"[{name}] batch == single-frame launches, bit for bit: {same}; max |diff| to the multi-pass chain {d}; faults {faults()}", flush=True)
    t_single = timed(lambda: [pipeline12_reinhard(f, out=o, whole_frame=True) for f, o in zip(fr, bp.outputs)], 30) / n
    t_batch = timed(lambda: bp(fr), 60) / n
    bg = BatchPipeline(n, H, W, dev, whole_frame=True, use_graph=True)
    bg.prepare(fr)
    t_graph = timed(lambda: bg(fr), 200) / n
    same_g = all(bool(torch.equal(bg.outputs[k], outs[k])) for k in range(n))
    print(f"[{name}] us per frame: launch by launch {t_single*1e6:.2f}, one launch per batch of {n} {t_batch*1e6:.2f}, graph replay {t_graph*1e6:.2f}; "
          f"graph == eager: {same_g}; faults {faults()}", flush=True)
    del bp, bg
if os.environ.get("ORACLE"):
    from oracle import c_oracle
    from tests.util import assert_close
    bp = BatchPipeline(n, H, W, dev, whole_frame=True)
    outs = bp(mixed)
    torch.cuda.synchronize()
    for k in range(min(n, 3)):
        src = (host_nu[k % len(host_nu)] if k % 3 == 1 else host[k % len(host)])
        ref = c_oracle.pipeline12_reinhard(src, work="f16", out="f16")
        print(f"frame {k} vs C oracle: max err {assert_close(outs[k].cpu().numpy(), ref, f'frame {k}')}", flush=True)
