"""First-light check of the whole-frame kernel: 4K frame against the C oracle, error flag, timing."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from taichi_image_amd import _native, synthetic, types
from taichi_image_amd.pipeline import pipeline12_reinhard
from oracle import c_oracle
from tests.util import assert_close
dev = torch.device("cuda", 0)
for (H, W) in ((64, 512), (3072, 4096)):
    packed = synthetic.synthetic_packed12(1, H, W)
    frame = torch.from_numpy(packed).to(dev)
    out = pipeline12_reinhard(frame, whole_frame=True)
    torch.cuda.synchronize()
    ws = _native.workspace(H, W, dev)
    off = int(_native.lib().mi_isp_workspace_error_offset(H, W))
    print(H, W, "error flag:", int(ws[off:off + 4].view(torch.int32).item()), flush=True)
    ref = c_oracle.pipeline12_reinhard(packed, work="f16", out="f16")
    fp = ws[:19 * 4].view(torch.float32).cpu().numpy()
    print("  FrameParams lo,hi,inv:", fp[0:3], "Bmin,Bmax,lmean,gmean:", fp[3:7], "mapkey,ei:", fp[10:12], "mean3", fp[12:15], "lo2,hi2,inv2", fp[15:18], flush=True)
    from oracle import isp_oracle as O
    rgb = O.bayer_to_rgb(O.decode12(packed, "f16", scaled=True))
    _, st = O.tonemap_reinhard(rgb, dtype="f16", return_intermediates=True)
    print("  oracle:", {k: (float(v) if np.ndim(v) == 0 else v) for k, v in st.items()}, flush=True)
    err = assert_close(out.cpu().numpy(), ref, f"mega {H}x{W}")
    print("  parity ok, max err", err, flush=True)
st = torch.cuda.current_stream(dev)
for _ in range(5): pipeline12_reinhard(frame, out=out, whole_frame=True)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(st)
for _ in range(50): pipeline12_reinhard(frame, out=out, whole_frame=True)
e1.record(st); e1.synchronize()
print(f"whole frame: {e0.elapsed_time(e1) * 1e3 / 50:.1f} us", flush=True)
