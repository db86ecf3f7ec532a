"""Whole-frame pipeline timed 60 times back to back while rocm-smi is sampled: is the occasional slow run a clock/power state?"""
import os, sys, time, subprocess, threading
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from taichi_image_amd import synthetic
from taichi_image_amd.pipeline import BatchPipeline
H, W = 3072, 4096
dev = torch.device("cuda", 0)
host = [synthetic.synthetic_packed12(i) for i in range(4)]
frames = [torch.from_numpy(host[i % 4]).to(dev) for i in range(8)]
kind = sys.argv[1] if len(sys.argv) > 1 else "whole"
bp = BatchPipeline(8, H, W, dev, use_graph=True, **({"whole_frame": True} if kind == "whole" else {"n_streams": 2}))
bp.prepare(frames)
stop = False
samples = []
def watch():
    while not stop:
        t = time.perf_counter()
        try:
            o = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower", "--showtemp", "--csv"], capture_output=True, text=True, timeout=10).stdout
        except Exception as e:
            o = repr(e)
        samples.append((t, o.strip().replace("\n", " | ")[:600]))
        time.sleep(0.05)
th = threading.Thread(target=watch); th.start()
t00 = time.perf_counter()
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    for _ in range(10): bp(frames)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(150): bp(frames)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f"run {i:2d} t={t0 - t00:7.3f}..{t1 - t00:7.3f}s  {(t1 - t0) / 1200 * 1e6:.1f} us/frame", flush=True)
stop = True; th.join()
for t, o in samples[:3] + samples[3::max(1, len(samples) // 25)]:
    print(f"smi t={t - t00:7.3f}s {o}")
