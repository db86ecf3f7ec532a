"""Why a whole-frame BatchPipeline created after other pipelines can run slower: replays bench.other_workloads' order
and prints the time of every instance, several timings per instance, with the addresses of its buffers."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from taichi_image_amd import synthetic, types
from taichi_image_amd.pipeline import BatchPipeline
from taichi_image_amd.synthetic import pack12
H, W = 3072, 4096
dev = torch.device("cuda", 0)
host = [synthetic.synthetic_packed12(i) for i in range(4)]
frames = [torch.from_numpy(host[i % 4]).to(dev) for i in range(8)]

def t(bp, fr, steps=150):
    for _ in range(10): bp(fr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): bp(fr)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (steps * 8) * 1e6

def run(tag, fr, reps=1, **kw):
    bp = BatchPipeline(8, H, W, dev, use_graph=True, **kw)
    bp.prepare(fr)
    us = [t(bp, fr) for _ in range(reps)]
    print(f"{tag}: " + " ".join(f"{u:.1f}" for u in us) + f" us/frame  ws={bp.ws.data_ptr():#x} out0={bp.outputs[0].data_ptr():#x} "
          f"out7={bp.outputs[7].data_ptr():#x}", flush=True)
    return bp

def rescale(p):
    b = p.reshape(H, -1, 3).astype(np.uint32)
    v = np.stack([b[..., 0] | ((b[..., 1] & 0xF) << 8), (b[..., 1] >> 4) | (b[..., 2] << 4)], -1).reshape(H, W)
    v = np.rint(v * 0.7 + 0.1 * 4095).astype(np.uint16)
    return pack12(v)

run("headline whole-frame", frames, 2, whole_frame=True)
run("u8 whole-frame", frames, 1, dtype=types.u8, whole_frame=True)
nonunit = [torch.from_numpy(rescale(host[i % 4])).to(dev) for i in range(8)]
run("multi-pass non-unit", nonunit, 1, n_streams=2)
run("multi-pass", frames, 1, n_streams=2)
run("whole-frame after", frames, 4, whole_frame=True)
run("whole-frame non-unit", nonunit, 2, whole_frame=True)
run("whole-frame again", frames, 3, whole_frame=True)
print(torch.cuda.memory_summary(dev, abbreviated=True)[:1500])
