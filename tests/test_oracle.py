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


# ---------------------------------------------------------------------------------------------
# Independent pins of the stages that otherwise rest on the restatement alone (round-2 verdict, item 8)
# ---------------------------------------------------------------------------------------------
def _published_filters():
    """The four 5x5 Malvar-He-Cutler filters (coefficients in eighths) per output colour and site kind."""
    g_at_rb = np.array([[0, 0, -1, 0, 0], [0, 0, 2, 0, 0], [-1, 2, 4, 2, -1], [0, 0, 2, 0, 0], [0, 0, -1, 0, 0]], float)
    row = np.array([[0, 0, 0.5, 0, 0], [0, -1, 0, -1, 0], [-1, 4, 5, 4, -1], [0, -1, 0, -1, 0], [0, 0, 0.5, 0, 0]], float)
    diag = np.array([[0, 0, -1.5, 0, 0], [0, 2, 0, 2, 0], [-1.5, 0, 6, 0, -1.5], [0, 2, 0, 2, 0], [0, 0, -1.5, 0, 0]], float)
    ident = np.zeros((5, 5)); ident[2, 2] = 8.0
    return g_at_rb, row, row.T.copy(), diag, ident


@pytest.mark.parametrize("shape", [(4, 4), (6, 8), (2, 6)])
def test_border_renormalisation_from_the_published_filters(rng, shape):
    """bayer.py:138-155 at the image frame: only in-bounds taps contribute and the sum is divided by the in-bounds
    weight sum.  Expected values are built here, pixel by pixel with plain loops, from the published 5x5 filters -
    not from the oracle's tap tables - for images in which EVERY pixel is a border pixel, all four patterns."""
    g_at_rb, row, col, diag, ident = _published_filters()
    H, W = shape
    cfa = rng.random((H, W)).astype(np.float32)
    red_site = {O.RGGB: (0, 0), O.GRBG: (0, 1), O.GBRG: (1, 0), O.BGGR: (1, 1)}
    for pattern, (pr, pc) in red_site.items():
        want = np.zeros((H, W, 3))
        for r in range(H):
            for c in range(W):
                is_r = (r % 2 == pr) and (c % 2 == pc)
                is_b = (r % 2 != pr) and (c % 2 != pc)
                g_r_row = (r % 2 == pr) and (c % 2 != pc)
                if is_r: fs = (ident, g_at_rb, diag)
                elif is_b: fs = (diag, g_at_rb, ident)
                elif g_r_row: fs = (row, ident, col)         # green with red left / right
                else: fs = (col, ident, row)                 # green with red above / below
                for ch, f in enumerate(fs):
                    num = den = 0.0
                    for dr in range(-2, 3):
                        for dc in range(-2, 3):
                            if 0 <= r + dr < H and 0 <= c + dc < W and f[dr + 2, dc + 2] != 0:
                                num += float(cfa[r + dr, c + dc]) * f[dr + 2, dc + 2]
                                den += f[dr + 2, dc + 2]
                    want[r, c, ch] = min(max(num / den, 0.0), 1.0)
        got = O.bayer_to_rgb(cfa, pattern).astype(np.float64)
        assert np.abs(got - want).max() < 3e-6, f"pattern {pattern} shape {shape}"


def test_bilinear_hand_computed_3x3_to_2x2():
    """interpolate.py:24-34,59-66 by hand: dst (r, c) samples p = (r / s, c / s) with s = 2/3 (no half-pixel offset),
    taps at trunc(p) and trunc(p) + 1 clamped to the edge, mix(x, y, a) = x (1 - a) + y a on rows, then columns."""
    src = np.array([[[0.0, 10.0, 100.0], [1.0, 11.0, 101.0], [2.0, 12.0, 102.0]],
                    [[3.0, 13.0, 103.0], [4.0, 14.0, 104.0], [5.0, 15.0, 105.0]],
                    [[6.0, 16.0, 106.0], [7.0, 17.0, 107.0], [8.0, 18.0, 108.0]]], np.float32)
    got = O.resize_bilinear(src, (2, 2), 2.0 / 3.0)
    # destination (0,0): p = (0, 0) -> src[0,0].  (0,1): p = (0, 1.5): halfway between columns 1 and 2 of row 0.
    # (1,0): p = (1.5, 0): halfway between rows 1 and 2 of column 0.  (1,1): the mean of the four lower-right pixels.
    want = np.array([[src[0, 0], (src[0, 1] + src[0, 2]) / 2],
                     [(src[1, 0] + src[2, 0]) / 2, (src[1, 1] + src[1, 2] + src[2, 1] + src[2, 2]) / 4]], np.float32)
    assert np.allclose(got, want, rtol=0, atol=1e-5), (got, want)
    # clamp-to-edge: upscaling 2x reads index 1 + 1 = 2 -> clamped to 1 at the last destination column
    up = O.resize_bilinear(src[:2, :2], (4, 4), 2.0)
    assert np.allclose(up[0, 3], (src[0, 1] + src[0, 1]) / 2) and np.allclose(up[0, 1], (src[0, 0] + src[0, 1]) / 2)


def _reinhard_devlin(rgb, key_bounds, log_mean, chan_mean, lum_mean, f_prime, a, c):
    """Reinhard & Devlin, "Dynamic range reduction inspired by photoreceptor physiology" (IEEE TVCG 2005), eqs. 1-7 in
    float64: V = I / (I + (f Ia)^m), Ia = a Il + (1 - a) Ig, Il = c Ic + (1 - c) L, Ig = c Cav + (1 - c) Lav,
    f = exp(-f'), m = 0.3 + 0.7 k^1.4, k = (Lmax - Lav) / (Lmax - Lmin) on log luminances."""
    lmin, lmax = key_bounds
    k = (lmax - log_mean) / (lmax - lmin)
    m = 0.3 + 0.7 * k ** 1.4
    f = np.exp(-f_prime)
    L = rgb @ np.array([0.299, 0.587, 0.114])
    Il = c * rgb + (1 - c) * L[..., None]
    Ig = c * np.asarray(chan_mean) + (1 - c) * lum_mean
    Ia = a * Il + (1 - a) * Ig
    return rgb / (rgb + (f * Ia) ** m)


@pytest.mark.parametrize("a,c,fp", [(1.0, 0.0, 1.0), (0.6, 0.3, 0.5), (0.0, 1.0, 2.0)])
def test_reinhard_matches_the_published_photoreceptor_model(a, c, fp):
    """camera_isp.py:186-218 and tonemap.py:108-131 against the published formula on a 2x2 image with hand-set
    statistics.  light_adapt = a, color_adapt = c, intensity = f'.  The stateless path keeps the reference's sign
    quirk (tonemap.py:102 stores -log_max): the published model with Lmax := -log_max, Lmin := log_min."""
    img = np.array([[[0.9, 0.5, 0.2], [0.05, 0.10, 0.02]], [[0.30, 0.30, 0.30], [0.6, 0.1, 0.7]]], np.float64)
    # ISP path: metering 9-vector = bounds (0, 1), log bounds, log mean, mean, rgb mean - set by hand
    m9 = np.array([0.0, 1.0, -4.0, -0.1, -1.5, 0.35, 0.4, 0.3, 0.25], np.float32)
    want = _reinhard_devlin(img, (m9[2], m9[3]), m9[4], m9[6:9], m9[5], fp, a, c)
    _, after = O.reinhard_isp(img.astype(np.float32), m9, gamma=1.0, intensity=fp, light_adapt=a, color_adapt=c)
    assert np.abs(after.astype(np.float64) - want).max() < 2e-6 * 4
    # and the u8 output of the second pass: 255 (p / max p)^(1 / gamma), truncated
    u8, _ = O.reinhard_isp(img.astype(np.float32), m9, gamma=0.6, intensity=fp, light_adapt=a, color_adapt=c)
    want8 = np.floor(255.0 * (want / max(1e-6, want.max())) ** (1 / 0.6))
    assert np.abs(u8.astype(np.float64) - want8).max() <= 1
    # stateless path: statistics computed by the oracle itself from the normalised image; rebuild the expectation
    # from those statistics with the published formula and the sign quirk
    out, st = O.tonemap_reinhard(img.astype(np.float32), 1.0, fp, a, c, "f32", return_intermediates=True)
    t = np.clip((img - float(st["lo"])) / (float(st["hi"]) - float(st["lo"])), 0, 1)
    g = t @ np.array([0.299, 0.587, 0.114])
    lg = np.log(np.maximum(g, 1e-4))
    assert abs(float(st["Bmin"]) - lg.min()) < 1e-6 and abs(float(st["Bmax"]) + lg.max()) < 1e-6   # (log_min, -log_max)
    v = _reinhard_devlin(t, (float(st["Bmin"]), float(st["Bmax"])), lg.mean(), t.reshape(-1, 3).mean(0), g.mean(), fp, a, c)
    want_out = (v - v.min()) / (v.max() - v.min())
    assert np.abs(out.astype(np.float64) - want_out).max() < 1e-5


def test_border_division_by_reciprocal():
    """csrc/isp_stream.h div16_by<T>: q = a RN(1/T), e = fma(-q, T, a), q' = fma(e, RN(1/T), q) equals
    the IEEE quotient a / T for every weight sum T that occurs at an image edge or corner (10..22; 16 is exact by
    construction).  oracle/check_recip_div.c checks EVERY finite float (2^32 patterns per T; MI_ISP_EXHAUSTIVE=1,
    ~2.5 min on 8 cores: all twelve came out clean on 2026-10-04); the default run
    takes every 97th bit pattern.  The only exceptions are numerators whose quotient is subnormal (|a| < 2^-120) and -0,
    which an accumulated pixel value cannot be (zero or at least 2^-29 in magnitude; at least one positive weight)."""
    import subprocess, tempfile
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "check_recip_div.c")
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "check")
        subprocess.run(["gcc", "-O2", "-march=x86-64-v3", "-fopenmp", "-ffp-contract=off", src, "-o", exe, "-lm"], check=True)
        stride = "1" if os.environ.get("MI_ISP_EXHAUSTIVE") else "97"
        out = subprocess.run([exe, stride], check=True, capture_output=True, text=True).stdout
    assert "FAIL" not in out and out.count("ok") == 12, out
