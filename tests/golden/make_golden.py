"""Regenerates tests/golden/golden_small.npz from the NumPy oracle.

The reference cannot run here (no taichi), so these vectors come from oracle/isp_oracle.py, which
is itself pinned by tests/golden/kat.json (hand-derived) and by the properties in
tests/test_oracle.py.  They serve two purposes: (1) freeze the oracle against accidental edits,
(2) let the GPU tests check the HIP path against committed data as well as against the live oracle.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import isp_oracle as O  # noqa: E402


def build():
    rng = np.random.default_rng(4242)
    g = {}
    enc = rng.integers(0, 256, 3 * 20).astype(np.uint8)
    g["dec_in"] = enc
    g["dec_std_u16"] = O.decode12(enc, "u16")
    g["dec_ids_u16"] = O.decode12(enc, "u16", ids_format=True)
    g["dec_std_f16s"] = O.decode12(enc, "f16", scaled=True)
    g["dec_ids_f32s"] = O.decode12(enc, "f32", scaled=True, ids_format=True)
    g["dec16_f16s"] = O.decode16(enc, "f16", scaled=True)

    cfa16 = rng.integers(0, 65536, (16, 16)).astype(np.uint16)
    cfah = rng.random((16, 16), dtype=np.float32).astype(np.float16)
    g["cfa_u16"], g["cfa_f16"] = cfa16, cfah
    ccm = O.isp_color_matrix(True, O.DEFAULT_WB, O.DEFAULT_CC)
    for p in range(4):
        g[f"rgb_u16_p{p}"] = O.bayer_to_rgb(cfa16, p)
        g[f"rgb_f16_p{p}"] = O.bayer_to_rgb(cfah, p)
    g["rgb_f16_ccm"] = O.bayer_to_rgb(cfah, 0, ccm)

    # 64 x 48 frame, the canonical call sequences (test/pipeline.py, test/camera_isp.py)
    H, W = 48, 64
    r = np.arange(H)[:, None] / H
    c = np.arange(W)[None, :] / W
    base = 0.1 + 0.8 * (0.5 + 0.5 * np.sin(6 * r + 1)) * (0.5 + 0.5 * np.cos(9 * c))
    img = np.stack([np.clip(base * k + rng.normal(0, 0.03, (H, W)), 0, 1) for k in (1.0, 0.8, 0.6)], -1).astype(np.float32)
    for p, ids in ((0, False), (2, True)):
        cfa = O.rgb_to_bayer(img, p)
        packed = O.encode12(cfa, scaled=True, ids_format=ids)
        tag = f"p{p}{'i' if ids else 's'}"
        g[f"packed_{tag}"] = packed
        g[f"pipe_f16_{tag}"] = O.pipeline12_reinhard(packed, p, ids)
        g[f"pipe_u8_{tag}"] = O.pipeline12_reinhard(packed, p, ids, out="u8", gamma=0.6)
    packed = g["packed_p0s"]
    im = O.isp_load_packed12(packed, "f16", resize_width=40)
    g["isp_img_f16"] = im
    st = O.IspState(0.1)
    m1 = st.update_metering([im, im])
    m2 = st.update_metering([im])
    g["isp_metrics_1"], g["isp_metrics_2"] = m1, m2
    u8, after = O.reinhard_isp(im, m2, gamma=0.6)
    g["isp_u8"], g["isp_after"] = u8, after
    g["isp_linear_u8"] = O.linear_isp(im, m2, 0.8)
    g["resize_f32"] = O.resize_bilinear(img, (30, 20), 0.46875)
    g["scene_f32"] = img                     # the source of resize_f32 (round 4: the GPU test reads the file alone)
    return g


if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_small.npz")
    np.savez_compressed(out, **build())
    print("wrote", out, os.path.getsize(out), "bytes")
