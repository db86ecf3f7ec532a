"""Phase timeline of the whole-frame kernel from in-kernel s_memtime stamps (lane 0 of every wave).
    make -C taichi_image_amd/csrc EXTRA="-DMI_STREAM_STAMPS -DMI_ISP_MEASURE" OBJDIR=../../build/csrc_stamps OUT=../lib/libmi355_isp_stamps.so
    MI_ISP_LIB=taichi_image_amd/lib/libmi355_isp_stamps.so python scripts/mega_stamps.py
"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from taichi_image_amd import _native, synthetic
from taichi_image_amd.pipeline import pipeline12_reinhard
H, W = 3072, 4096
dev = torch.device("cuda", 0)
host = synthetic.synthetic_packed12(0)
if os.environ.get("NONUNIT"):          # scene scaled into [0.1, 0.8]: bounds other than (0, 1), phase B and its barrier run
    b = host.reshape(H, -1, 3).astype(np.uint32)
    v = np.stack([b[..., 0] | ((b[..., 1] & 0xF) << 8), (b[..., 1] >> 4) | (b[..., 2] << 4)], -1).reshape(H, W)
    host = synthetic.pack12(np.rint(v * 0.7 + 0.1 * 4095).astype(np.uint16))
frame = torch.from_numpy(host).to(dev)
out = torch.empty((H, W, 3), dtype=torch.float16, device=dev)
for _ in range(5): pipeline12_reinhard(frame, out=out, whole_frame=True)
torch.cuda.synchronize()
ws = _native.workspace(H, W, dev)
ws[(64 + 48 * 4096) * 4:].zero_()            # the stamp area only (the sync words must stay as the kernel left them)
for _ in range(int(os.environ.get('N_B2B', '1'))): pipeline12_reinhard(frame, out=out, whole_frame=True)   # the stamps of the last launch stay
torch.cuda.synchronize()
raw = ws.cpu().numpy().view(np.uint32)
base = 64 + 48 * 4096
nw = 2048
s = raw[base:base + nw * 16].reshape(nw, 16).astype(np.int64)
names = ["entry", "phase A done", "reduced+signalled", "barrier 0 + bounds", "stats folded", "phase C done", "reduced+signalled",
         "barrier 2 + bounds2", "phase D done"]
rel = (s - s[:, :1]) & 0xFFFFFFFF
print(f"{'phase':24s}   mean    p10    p50    p90   [ticks since the previous stamp]")
for i in range(1, 9):
    d = rel[:, i] - rel[:, i - 1]
    print(f"{names[i]:24s} {d.mean():7.0f} {np.percentile(d,10):6.0f} {np.percentile(d,50):6.0f} {np.percentile(d,90):6.0f}")
print("wave lifetime", rel[:, 8].mean(), np.percentile(rel[:, 8], [10, 50, 90]))

# the folding wave of each block: stamps 9..11 (barrier 0) and 12..14 (barrier 2): poll done, loads done, scalars published
for name, b0, ref in (("barrier 0", 9, 2), ("barrier 2", 12, 6)):
    f = s[:, b0] != 0
    if f.any():
        d = (s[f][:, [b0, b0 + 1]] - s[f][:, [ref]]) & 0xFFFFFFFF
        print(name, "role-0 waves:", int(f.sum()), " poll matched / partial fold left in LDS, ticks after the wave's own arrival: mean",
              d.mean(axis=0).astype(int), "p90", np.percentile(d, 90, axis=0).astype(int))

# absolute timeline: needs the stamps library built with -DMI_STAMP_REALTIME (100 MHz counter shared by the chip);
# with the per-CU cycle counter the rows below are meaningless
if os.environ.get("STAMP_RT"):
    t0 = s[:, 0].min()
    ab = (s[:, :9] - t0) / 100.0
    print(f"\n{'absolute [us]':24s}    min    p10    p50    p90    max")
    for i in range(9):
        a = ab[:, i]
        print(f"{names[i] if i else 'entry':24s} {a.min():7.2f} {np.percentile(a,10):6.2f} {np.percentile(a,50):6.2f} {np.percentile(a,90):6.2f} {a.max():6.2f}")
    for name, b0 in (("barrier 0", 9), ("barrier 2", 12)):
        f = s[:, b0] != 0
        d = (s[f][:, [b0, b0 + 1]] - t0) / 100.0
        print(name, "role-0 waves: poll matched min/p50/max", np.round(np.percentile(d[:, 0], [0, 50, 100]), 2),
              " partial fold left in LDS", np.round(np.percentile(d[:, 1], [0, 50, 100]), 2))
    order = np.argsort(ab[:, 2])
    print("last arrivals at barrier 0 (wave, block, entry, A done, signalled):", [(int(w), int(w // 4), round(ab[w, 0], 2), round(ab[w, 1], 2), round(ab[w, 2], 2)) for w in order[-6:]])
    order = np.argsort(ab[:, 6])
    print("last arrivals at barrier 2 (wave, block, C done, signalled):", [(int(w), int(w // 4), round(ab[w, 5], 2), round(ab[w, 6], 2)) for w in order[-6:]])
if os.environ.get("STAMP_RT"):
    dur = (ab[:, 1] - ab[:, 0]).reshape(512, 4)
    print("phase A duration per block [us] (max over its waves): blocks 0,1,2,3:", np.round(dur[:4].max(axis=1), 2), " 508..511:", np.round(dur[508:].max(axis=1), 2),
          " all blocks p50/p90/max:", np.round(np.percentile(dur.max(axis=1), [50, 90, 100]), 2))
    slow = np.argsort(dur.max(axis=1))[-8:]
    print("slowest phase-A blocks:", [(int(b), round(float(dur[b].max()), 2)) for b in slow])
    durc = (ab[:, 5] - ab[:, 4]).reshape(512, 4)
    print("phase C duration per block: p50/p90/max", np.round(np.percentile(durc.max(axis=1), [50, 90, 100]), 2), " slowest:", [(int(b), round(float(durc[b].max()), 2)) for b in np.argsort(durc.max(axis=1))[-6:]])
    sig = (ab[:, 2] - ab[:, 1])
    print("reduce+post after phase A: p50/p90/p99/max", np.round(np.percentile(sig, [50, 90, 99, 100]), 2), " slowest waves:", [(int(w), int(w // 4), round(float(sig[w]), 2)) for w in np.argsort(sig)[-6:]])
    if os.environ.get("STAMP_HWID"):          # stamps build with -DMI_STAMP_HWID: slots 14 / 15 hold XCC_ID / HW_ID
        hw = s[:, 15].astype(np.int64); xcc = (np.arange(2048) // 4) % 8
        cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7; simd = (hw >> 4) & 3
        place = (xcc << 12 | se << 8 | sh << 4 | cu)
        pb = place.reshape(512, 4)
        from collections import Counter
        print("waves of a block on one CU:", bool((pb == pb[:, :1]).all()), " distinct CUs:", len(set(pb[:, 0].tolist())), " blocks per CU:", Counter(Counter(pb[:, 0].tolist()).values()))
        for b in (0, 1, 254, 255, 256, 257, 510, 511):
            mates = [int(x) for x in np.nonzero(pb[:, 0] == pb[b, 0])[0]]
            print(f"block {b}: xcc {int(xcc[4*b])} se {int(se[4*b])} sh {int(sh[4*b])} cu {int(cu[4*b])} simds {simd[4*b:4*b+4].tolist()}  blocks on this CU: {mates}  phase A {dur[b].max():.2f} us")
    else:                                     # slots 15 / 14: prologue done, half of phase A's rows done
        pro = ((s[:, 15] - s[:, 0]) / 100.0).reshape(512, 4); half = ((s[:, 14] - s[:, 15]) / 100.0).reshape(512, 4)
        rest = ((s[:, 1] - s[:, 14]) / 100.0).reshape(512, 4)
        med = lambda x: np.median(x, axis=1)
        print("phase A split, median over blocks of the per-block median wave [us]: prologue", round(float(np.median(med(pro))), 2), " rows 0-5", round(float(np.median(med(half))), 2), " rows 6-11", round(float(np.median(med(rest))), 2))
        for b in (0, 1, 2, 255, 256, 257, 258, 509, 510, 511):
            print(f"  block {b}: prologue {med(pro)[b]:.2f}  rows 0-5 {med(half)[b]:.2f}  rows 6-11 {med(rest)[b]:.2f}")
