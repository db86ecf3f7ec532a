"""Does a captured hipGraph of one 8-frame step (2 streams, 32 kernel nodes) replay faster than issuing it?"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from taichi_image_amd import synthetic
from taichi_image_amd.pipeline import BatchPipeline
H, W = 3072, 4096
dev = torch.device("cuda", 0)
for n_streams in (1, 2, 3):
    frames = [torch.from_numpy(synthetic.synthetic_packed12(i % 2)).to(dev) for i in range(8)]
    bp = BatchPipeline(8, H, W, dev, n_streams=n_streams, whole_frame=False)
    for _ in range(3): bp(frames)
    torch.cuda.synchronize()
    def timeit(fn, n=30):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n / 8 * 1e6
    eager = timeit(lambda: bp(frames))
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        bp(frames)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            bp(frames)
    torch.cuda.synchronize()
    graph = timeit(g.replay)
    print(f"streams={n_streams}: eager {eager:.1f} us/frame, graph replay {graph:.1f} us/frame", flush=True)
