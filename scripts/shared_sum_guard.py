"""Would an exactness-guarded shared-sum demosaic (verdict r2 item 1c) ever take its fast path?

Sharing partial sums between the channels' 13-tap filters (bayer.py:30-55) reorders fp32 additions; the result keeps the
reference's bits only if every partial sum is exact.  Inputs are f16 values x = m * 2^(e-10) (11-bit m), weights w/16 with
|w| <= 16: all products are multiples of 2^(e_min-14) and every partial sum is below 2^(e_max+2), so 24 bits suffice iff
e_max - e_min <= 8 over the NON-ZERO values of the window.  A branch must be wave-uniform: the window of a wave row is
512 columns (+2 either side) x 6 rows.  This script evaluates the guard on the benchmark's own frames
- per pixel (5x5 window), per lane (12 columns x 6 rows) and per wave row - CPU only.
    python scripts/shared_sum_guard.py [n_frames] > profiles/r03_shared_sum_guard.txt"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from taichi_image_amd import synthetic          # host-side generator only (numpy)

def unpack12(p, H, W):
    b = p.reshape(H, -1, 3).astype(np.uint32)
    return np.stack([b[..., 0] | ((b[..., 1] & 0xF) << 8), (b[..., 1] >> 4) | (b[..., 2] << 4)], -1).reshape(H, W)

def guard_fraction(v12, rows, cols):
    """fraction of windows of rows x cols (stepped by rows-4 / cols-4: the interior a window serves) whose non-zero f16
    exponents span at most 8"""
    x = (v12.astype(np.float32) * np.float32(1.0 / 4095.0)).astype(np.float16)
    e = np.frexp(x.astype(np.float32))[1].astype(np.int32)          # exponent of the value (0 for 0)
    nz = x != 0
    emax = np.where(nz, e, -100); emin = np.where(nz, e, 100)
    H, W = x.shape
    ok = tot = 0
    sr, sc = max(rows - 4, 1), max(cols - 4, 1)
    for r0 in range(0, H - rows + 1, sr):
        a = emax[r0:r0 + rows].max(axis=0); b = emin[r0:r0 + rows].min(axis=0)
        # sliding window over columns via strided max / min
        n = (W - cols) // sc + 1
        idx = np.arange(n)[:, None] * sc + np.arange(cols)[None, :]
        hi = a[idx].max(axis=1); lo = b[idx].min(axis=1)
        ok += int(((hi - lo <= 8) | (hi < -50)).sum()); tot += n
    return ok / tot

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
print("# scripts/shared_sum_guard.py: fraction of windows in which a reordered (shared-sum) fp32 demosaic is provably exact")
print("# (non-zero f16 values of the window span at most 8 binades), SURVEY 8(d) synthetic frames, 4096 x 3072")
for k in range(n):
    v12 = unpack12(synthetic.synthetic_packed12(k), 3072, 4096)
    print(f"frame {k}: per pixel (5 x 5) {guard_fraction(v12, 5, 5):.4f}   per lane (6 rows x 12 columns) {guard_fraction(v12, 6, 12):.4f}   "
          f"per wave row (6 rows x 516 columns) {guard_fraction(v12, 6, 516):.4f}   zeros {float((v12 == 0).mean()):.4f}", flush=True)
