"""CPU tests of the oracle itself: hand-derived known answers, the reference's only assertion
(test/packed.py:6-15), algebraic properties and the committed golden vectors.  No GPU, no HIP."""
import json
import os

import numpy as np
import pytest

from oracle import isp_oracle as O
from tests.util import assert_exact

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "kat.json")))


def test_kat_unpack():
    a = KAT["decode12_standard"]
    assert O.decode12(np.array(a["bytes"], np.uint8)).tolist() == a["values"]
    b = KAT["decode12_ids"]
    assert O.decode12(np.array(b["bytes"], np.uint8), ids_format=True).tolist() == b["values"]
    c = KAT["decode16"]
    assert O.decode16(np.array(c["bytes"], np.uint8)).tolist() == c["values"]
    s = KAT["scaled_f16_bits"]
    v = np.array(s["values"] + [0] * (len(s["values"]) % 2), np.uint16)
    d = O.decode12(O.encode12(v), "f16", scaled=True).view(np.uint16)[: len(s["values"])]
    assert d.tolist() == s["bits"]


def test_kat_bayer_tables():
    assert [list(t) for t in O.DIAMOND] == KAT["tap_order"]
    assert O.BAYER_KERNELS.tolist() == KAT["bayer_kernels"]
    assert np.all(O.BAYER_KERNELS.sum(axis=1) == 16)
    for k, v in KAT["kernel_patterns"].items():
        assert list(O.KERNEL_PATTERN[int(k)]) == v
    for k, v in KAT["pixel_orders"].items():
        assert list(O.PIXEL_ORDER[int(k)]) == v


def test_reference_roundtrip(rng):
    """taichi_image/test/packed.py:6-15."""
    for _ in range(100):
        size = int(rng.integers(0, 1000)) * 2
        x = rng.integers(0, 2 ** 12, size=size).astype(np.uint16)
        assert np.all(O.decode12(O.encode12(x)) == x)


def test_scaled_roundtrip_is_identity_on_12bit_grid(rng):
    v = rng.integers(0, 4096, 2000).astype(np.uint16)
    for dt in ("f32", "u16"):
        x = O.decode12(O.encode12(v), dt, scaled=True)
        assert np.array_equal(O.decode12(O.encode12(x, scaled=True)), v), dt


@pytest.mark.parametrize("p", [0, 1, 2, 3])
@pytest.mark.parametrize("dtype,val", [("u8", 200), ("u16", 1000), ("f16", 0.37), ("f32", 0.123)])
def test_constant_cfa_invariance(p, dtype, val):
    """Channel weights sum to 16 over any in-bounds tap subset the borders produce, so a
    constant CFA stays constant.  (Integer outputs may drop one LSB: truncation of v/s*s.)"""
    cfa = np.full((12, 10), val, O.NP_DTYPE[dtype])
    rgb = O.bayer_to_rgb(cfa, p)
    if dtype == "f16":      # k*v is exact in fp32 for every partial sum -> exactly constant
        assert np.all(rgb == cfa[0, 0])
    elif dtype == "f32":    # products round: constant to within an ulp or two
        assert np.allclose(rgb, cfa[0, 0], rtol=5e-7, atol=0)
    else:
        assert np.all((rgb.astype(int) - int(val)) >= -1) and np.all(rgb <= val)
        assert len(np.unique(rgb)) == 1


def test_demosaic_site_passthrough(rng):
    """The identity kernel: each pixel's own colour passes through untouched (interior)."""
    for p in range(4):
        cfa = rng.random((16, 20), dtype=np.float32)
        rgb = O.bayer_to_rgb(cfa, p)
        order = O.PIXEL_ORDER[p]
        for site, (i, k) in enumerate([(0, 0), (0, 1), (1, 0), (1, 1)]):
            ch = order[site]
            assert np.array_equal(rgb[i::2, k::2, ch], np.clip(cfa[i::2, k::2], 0, 1)), (p, site)


def test_mosaic_demosaic_psnr(rng):
    """Workflow of taichi_image/test/bayer.py:56-65 on a smooth scene, all four patterns."""
    r = np.arange(64)[:, None] / 64
    c = np.arange(96)[None, :] / 96
    img = np.stack([0.5 + 0.4 * np.sin(5 * r + k) * np.cos(4 * c) for k in range(3)], -1).astype(np.float32)
    for p in range(4):
        out = O.bayer_to_rgb(O.rgb_to_bayer(img, p), p)
        mse = np.mean((out[4:-4, 4:-4] - img[4:-4, 4:-4]) ** 2)
        assert 10 * np.log10(1.0 / mse) > 45, p


def test_resize_identity_and_clamp(rng):
    src = rng.random((9, 7, 3), dtype=np.float32)
    assert_exact(O.resize_bilinear(src, (7, 9), 1.0), src)
    up = O.resize_bilinear(src, (14, 18), 2.0)
    assert_exact(up[::2, ::2], src)
    assert np.array_equal(up[17, ::2], src[8])          # clamp-to-edge: last half-row repeats the edge


def test_isp_output_size_rounding():
    assert O.isp_output_size(3072, 4096, 1920) == ((1920, 1440), 1920 / 4096)
    assert O.isp_output_size(5, 10, scale=0.5)[0] == (5, 2)     # round(2.5) == 2 (banker's)


def test_transform_identities(rng):
    x = (rng.random((6, 10, 3)) * 255).astype(np.uint8)
    T = O.transform
    assert np.array_equal(T(T(x, "rotate_90"), "rotate_270"), x)
    assert np.array_equal(T(T(x, "rotate_180"), "rotate_180"), x)
    assert np.array_equal(T(x, "transpose"), x.transpose(1, 0, 2))
    assert np.array_equal(T(x, "flip_vert"), x[::-1])
    assert np.array_equal(T(x, "flip_horiz"), x[:, ::-1])
    assert np.array_equal(T(x, "rotate_90"), np.rot90(x, -1))
    sq = x[:6, :6]
    assert np.array_equal(T(T(sq, "transverse"), "transverse"), sq)
    with pytest.raises(AssertionError):
        T(x, "transverse")


def test_metering_bounds_blended_twice(rng):
    """camera_isp.py:156-157 then :165: the new bounds enter with weight (1-alpha)^2."""
    im = rng.random((32, 32, 3), dtype=np.float32).astype(np.float16)
    prev = np.array([0.25, 0.5, -3, -0.1, -1, 0.4, 0.4, 0.4, 0.4], np.float32)
    a = 0.7
    m = O.metering_images([im], a, prev)
    s = im[::8, ::8].astype(np.float32)
    lo, hi = s.min(), s.max()
    assert np.isclose(m[0], (1 - a) ** 2 * lo + (1 - (1 - a) ** 2) * prev[0], rtol=1e-6)
    assert np.isclose(m[1], (1 - a) ** 2 * hi + (1 - (1 - a) ** 2) * prev[1], rtol=1e-6)
    first = O.metering_images([im], 0.0, np.zeros(9, np.float32))
    assert first[0] == lo and first[1] == hi


def test_sharded_metering_equals_whole(rng):
    """The split used for multi-GPU: partials of two shards combine to the single-shard result."""
    ims = [rng.random((40, 48, 3), dtype=np.float32).astype(np.float16) for _ in range(4)]
    prev = np.array([0.1, 0.9, -3, -0.1, -1, 0.4, 0.4, 0.4, 0.4], np.float32)
    a = 0.9
    whole = O.metering_images(ims, a, prev)
    b0, b1 = O.metering_partials_bounds(ims[:1]), O.metering_partials_bounds(ims[1:])
    raw = np.array([min(b0[0], b1[0]), max(b0[1], b1[1])], np.float32)
    b = (raw + np.float32(a) * (prev[:2] - raw)).astype(np.float32)
    (p0, n0), (p1, n1) = O.metering_partials_sums(ims[:1], b), O.metering_partials_sums(ims[1:], b)
    part = np.array([min(p0[0], p1[0]), max(p0[1], p1[1]), *(p0[2:] + p1[2:])])
    got = O.metering_finish(prev, b, part, n0 + n1, a)
    assert np.allclose(got, whole, rtol=1e-6, atol=1e-7)


def test_stateless_reinhard_range_and_quirk(rng):
    img = rng.random((24, 32, 3), dtype=np.float32).astype(np.float16)
    out, info = O.tonemap_reinhard(img, dtype="f32", return_intermediates=True)
    assert out.min() == 0.0 and out.max() == 1.0
    assert info["Bmax"] >= 0 and info["Bmin"] <= 0          # B = (log_min, -log_max) sign quirk
    out8 = O.tonemap_reinhard(img, dtype="u8")
    assert out8.dtype == np.uint8 and out8.max() == 255


def test_golden_vectors_frozen():
    """The committed fixtures equal what the oracle produces today (see make_golden.py)."""
    from tests.golden.make_golden import build
    gold = np.load(os.path.join(HERE, "golden", "golden_small.npz"))
    now = build()
    assert sorted(gold.files) == sorted(now)
    for k in gold.files:
        assert_exact(now[k], gold[k], k)


# ---- color/yuv_420.py --------------------------------------------------------------------------
def test_yuv420_known_answers():
    """Hand-derived from yuv_420.py:12-27,52-59: the BGR-named matrix sees rgb.bgr, planes are
    (V, U) in that order, float -> u8 truncates."""
    white = np.full((2, 2, 3), 255, np.uint8)
    yuv = O.rgb_yuv420(white)
    assert yuv.shape == (3, 2) and yuv.tolist() == [[255, 255], [255, 255], [127, 127]]
    red = np.zeros((2, 2, 3), np.uint8); red[..., 0] = 255
    # (b, g, r) = (0, 0, 1): Y = 0.114, yuv.y = 0.5 + 0.5 = 1.0 (plane 1), yuv.z = 0.5 - 0.081312 (plane 0)
    assert O.rgb_yuv420(red).tolist() == [[29, 29], [29, 29], [106, 255]]
    y, uv, (w, h) = O.split_yuv_420(np.zeros((12, 8), np.uint8))
    assert y.shape == (8, 8) and uv.shape == (2, 4, 4) and (w, h) == (8, 8)


def test_yuv420_flat_round_trip_and_clamp_quirk(rng):
    flat = np.empty((4, 6, 3), np.float32)
    flat[...] = [0.2, 0.5, 0.7]
    back = O.yuv420_rgb(O.rgb_yuv420(flat))
    assert np.abs(back - flat).max() < 1e-5            # lossless on a constant image (fp32)
    # tm.clamp(0, 1, x) == min(1, x): values above 1 saturate, negative ones pass through
    hot = np.full((2, 2, 3), 1.5, np.float32)
    assert O.rgb_yuv420(hot)[:2].max() == 1.0
    yuv = np.zeros((3, 2), np.float32); yuv[2] = [0.0, 0.0]     # y = 0, u = v = -0.5 after the offset
    assert O.yuv420_rgb(yuv).min() < 0.0


def test_scaled_division_trick():
    """csrc/isp_common.h div_scale: x * RN(1/d) with one FMA residual correction equals the IEEE
    division x / d for every integer x in [0, d], d = 255 and 65535 (the plain product does not)."""
    f32 = np.float32
    for d in (255.0, 65535.0):
        x = np.arange(0, int(d) + 1, dtype=np.float64)
        r = f32(1.0) / f32(d)
        q = (x.astype(f32) * r).astype(f32)
        e = (x - q.astype(np.float64) * d).astype(f32)                  # fma(-q, d, x): exact product, one rounding
        q2 = (q.astype(np.float64) + e.astype(np.float64) * np.float64(r)).astype(f32)
        want = x.astype(f32) / f32(d)
        assert (q != want).any() and np.array_equal(q2, want)


def test_demosaic_equals_published_malvar_he_cutler(rng):
    """An independent pin of the weight tables: the reference's kernels are Malvar-He-Cutler 2004 ("High-quality
    linear interpolation for demosaicing of Bayer-patterned color images", ICASSP 2004; the reference's own
    test/compare_bayer.py compares against that algorithm).  The paper's four 5x5 filters (coefficients / 8),
    applied with plain correlations, reproduce the oracle's interior for all four CFA patterns."""
    from scipy.ndimage import correlate
    g_at_rb = np.array([[0, 0, -1, 0, 0], [0, 0, 2, 0, 0], [-1, 2, 4, 2, -1], [0, 0, 2, 0, 0], [0, 0, -1, 0, 0]]) / 8.0
    # R at a green site whose row holds R (horizontal R neighbours); its transpose serves the other green site
    rb_at_g_row = np.array([[0, 0, 0.5, 0, 0], [0, -1, 0, -1, 0], [-1, 4, 5, 4, -1], [0, -1, 0, -1, 0], [0, 0, 0.5, 0, 0]]) / 8.0
    rb_at_g_col = rb_at_g_row.T
    rb_at_br = np.array([[0, 0, -1.5, 0, 0], [0, 2, 0, 2, 0], [-1.5, 0, 6, 0, -1.5], [0, 2, 0, 2, 0], [0, 0, -1.5, 0, 0]]) / 8.0
    H, W = 24, 32
    cfa = rng.random((H, W)).astype(np.float64)
    conv = {k: correlate(cfa, f, mode="constant") for k, f in
            (("g", g_at_rb), ("row", rb_at_g_row), ("col", rb_at_g_col), ("diag", rb_at_br))}
    rr, cc = np.mgrid[0:H, 0:W]
    # (row parity, col parity) of the red site per pattern (bayer.py:85-90 pixel orders)
    red_site = {O.RGGB: (0, 0), O.GRBG: (0, 1), O.GBRG: (1, 0), O.BGGR: (1, 1)}
    for pattern, (pr, pc) in red_site.items():
        is_r = ((rr % 2) == pr) & ((cc % 2) == pc)
        is_b = ((rr % 2) != pr) & ((cc % 2) != pc)
        g_in_r_row = ((rr % 2) == pr) & ((cc % 2) != pc)          # green with red left / right
        g_in_b_row = ((rr % 2) != pr) & ((cc % 2) == pc)          # green with red above / below
        R = np.where(is_r, cfa, np.where(g_in_r_row, conv["row"], np.where(g_in_b_row, conv["col"], conv["diag"])))
        G = np.where(is_r | is_b, conv["g"], cfa)
        B = np.where(is_b, cfa, np.where(g_in_b_row, conv["row"], np.where(g_in_r_row, conv["col"], conv["diag"])))
        want = np.clip(np.stack([R, G, B], -1), 0, 1)
        got = O.bayer_to_rgb(cfa.astype(np.float32), pattern).astype(np.float64)
        assert np.abs(got[2:-2, 2:-2] - want[2:-2, 2:-2]).max() < 2e-6, f"pattern {pattern}"
