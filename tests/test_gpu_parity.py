"""GPU parity tests: the HIP path (through the C ABI, via the Python call surface that mirrors the
reference) against the CPU oracle on the same seeded inputs.

Contract (BASELINE.json north_star): bit-exact for unpack / integer indexing -- and, because the
kernels keep the reference's operation order, also for demosaic and bilinear resize; within
1e-4 relative (+1 unit of the output type) for the transcendental tonemap stages.
"""
import numpy as np
import pytest
import torch

from oracle import isp_oracle as O
from tests.util import reuse_case, assert_close, assert_exact, natural_packed12, random_cfa

pytestmark = pytest.mark.gpu

DT = ["u8", "u16", "f16", "f32"]


@pytest.fixture(scope="module")
def ti():
    import taichi_image_amd as t
    return t


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda", 0)


def tok(ti, name):
    return getattr(ti.types, name)


def pat(ti, p):
    return ti.BayerPattern(p)


# ---------------------------------------------------------------------------------------------
# packed.py
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ids", [False, True])
@pytest.mark.parametrize("dtype,scaled", [("u16", False), ("u16", True), ("f16", True), ("f32", True), ("f16", False),
                                          ("u8", True)])
@pytest.mark.parametrize("n_pairs", [0, 1, 3, 4, 37, 4096 + 5])
def test_decode12(ti, rng, ids, dtype, scaled, n_pairs):
    enc = rng.integers(0, 256, n_pairs * 3).astype(np.uint8)
    ref = O.decode12(enc, dtype, scaled, ids)
    got = ti.packed.decode12(enc, tok(ti, dtype), scaled=scaled, ids_format=ids)
    assert_exact(got, ref, f"decode12 {dtype} scaled={scaled} ids={ids}")


def test_decode12_kat(ti):
    """Hand-derived from packed.py:24-31,37-44."""
    enc = np.array([0xAB, 0xCD, 0xEF], np.uint8)
    assert ti.packed.decode12(enc).tolist() == [0xDAB, 0xEFC]
    assert ti.packed.decode12(enc, ids_format=True).tolist() == [0xABF, 0xCDE]
    v = np.array([0, 1, 2047, 2048, 3499, 3836, 4095, 0], np.uint16)
    d = ti.packed.decode12(ti.packed.encode12(v), ti.types.f16, scaled=True)
    assert d.view(np.uint16).tolist() == [0x0000, 0x0C00, 0x3800, 0x3800, 0x3AD6, 0x3B7E, 0x3C00, 0x0000]
    assert ti.packed.decode16(np.array([0x34, 0x12], np.uint8)).tolist() == [0x1234]


def test_decode12_shapes_and_unaligned(ti, rng, dev):
    enc = rng.integers(0, 256, (6, 5, 24)).astype(np.uint8)
    assert_exact(ti.packed.decode12(enc, ti.types.f32, scaled=True), O.decode12(enc, "f32", True))
    # a device view that is not 4-byte aligned exercises the byte path
    buf = torch.from_numpy(rng.integers(0, 256, 3 * 1001 + 1).astype(np.uint8)).to(dev)
    view = buf[1:]
    got = ti.packed.decode12(view, ti.types.u16)
    assert got.device == view.device and got.dtype == torch.uint16
    assert_exact(got.cpu().numpy(), O.decode12(buf.cpu().numpy()[1:], "u16"))


def test_encode_decode_roundtrip(ti, rng):
    """The reference's only assertion (test/packed.py:6-15), plus the scaled variants."""
    for _ in range(20):
        size = int(rng.integers(0, 1000)) * 2
        x = rng.integers(0, 2 ** 12, size=size).astype(np.uint16)
        assert np.all(ti.packed.decode12(ti.packed.encode12(x)) == x)
    x = rng.integers(0, 2 ** 12, size=(64, 258)).astype(np.uint16)
    for ids in (False, True):
        assert_exact(ti.packed.encode12(x, ids_format=ids), O.encode12(x, ids_format=ids))
    xf = rng.random((32, 64), dtype=np.float32)
    for dt in ("f32", "f16"):
        xs = xf.astype(O.NP_DTYPE[dt])
        assert_exact(ti.packed.encode12(xs, scaled=True), O.encode12(xs, scaled=True), f"encode12 scaled {dt}")
    x16 = rng.integers(0, 65536, size=(8, 32)).astype(np.uint16)
    assert_exact(ti.packed.encode12(x16, scaled=True), O.encode12(x16, scaled=True), "encode12 scaled u16")


def test_ids_layout_is_not_self_inverse(ti, rng):
    """Reference quirk reproduced: packed.py:48-55 puts p0's low nibble in the HIGH nibble of byte 2,
    packed.py:37-44 reads it from the LOW nibble, so IDS encode->decode swaps the low nibbles."""
    x = rng.integers(0, 4096, 64).astype(np.uint16)
    d = ti.packed.decode12(ti.packed.encode12(x, ids_format=True), ids_format=True)
    p0, p1 = x[0::2], x[1::2]
    assert np.array_equal(d[0::2], (p0 & 0xFF0) | (p1 & 0xF))
    assert np.array_equal(d[1::2], (p1 & 0xFF0) | (p0 & 0xF))


@pytest.mark.parametrize("dtype,scaled", [("u16", False), ("f16", True), ("f32", True), ("u8", True)])
@pytest.mark.parametrize("n", [0, 1, 7, 8, 1000 + 3])
def test_decode16(ti, rng, dtype, scaled, n):
    enc = rng.integers(0, 256, n * 2).astype(np.uint8)
    assert_exact(ti.packed.decode16(enc, tok(ti, dtype), scaled=scaled), O.decode16(enc, dtype, scaled))


# ---------------------------------------------------------------------------------------------
# bayer.py
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("p", [0, 1, 2, 3])
def test_rgb_to_bayer(ti, rng, p):
    img = rng.random((10, 14, 3), dtype=np.float32)
    assert_exact(ti.bayer.rgb_to_bayer(img, pat(ti, p)), O.rgb_to_bayer(img, p))
    img8 = rng.integers(0, 256, (4, 6, 3)).astype(np.uint8)
    assert_exact(ti.bayer.rgb_to_bayer(img8, pat(ti, p)), O.rgb_to_bayer(img8, p))


@pytest.mark.parametrize("p", [0, 1, 2, 3])
@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("shape", [(2, 2), (4, 6), (34, 130), (66, 264), (64, 256), (40, 128)])
def test_demosaic_bit_exact(ti, rng, p, dtype, shape):
    cfa = random_cfa(rng, *shape, dtype)
    got = ti.bayer.bayer_to_rgb(cfa, pat(ti, p))
    assert_exact(got, O.bayer_to_rgb(cfa, p), f"demosaic {dtype} pattern {p} {shape}")


@pytest.mark.parametrize("din,dout", [("f16", "f32"), ("f32", "f16"), ("u16", "f16"), ("u8", "f32"), ("f16", "u8"),
                                      ("u16", "u8"), ("f32", "u16"), ("u8", "u16")])
def test_demosaic_cross_dtype_and_ccm(ti, rng, din, dout):
    cfa = random_cfa(rng, 70, 150, din)
    assert_exact(ti.bayer.bayer_to_rgb(cfa, dtype=tok(ti, dout)), O.bayer_to_rgb(cfa, dtype=dout))
    ccm = O.isp_color_matrix(True, O.DEFAULT_WB, O.DEFAULT_CC)
    got = ti.bayer.bayer_to_rgb(cfa, pat(ti, 2), correct_colors=ccm, dtype=tok(ti, dout))
    assert_exact(got, O.bayer_to_rgb(cfa, 2, ccm, dout), f"ccm {din}->{dout}")


def test_demosaic_config1_full(ti):
    """BASELINE config 1: 1920x1080 RGGB 16-bit frame, seed 1."""
    cfa = np.random.default_rng(1).integers(0, 65536, (1080, 1920)).astype(np.uint16)
    got = ti.bayer.bayer_to_rgb(cfa)
    assert got.dtype == np.uint16 and got.shape == (1080, 1920, 3)
    assert_exact(got, O.bayer_to_rgb(cfa), "config 1")


def test_demosaic_constant_invariance_full_size(ti, dev):
    """Every kernel's channel weights sum to 16 (also over the in-bounds taps at the borders),
    so a constant CFA demosaics to the same constant everywhere -- checked at 4096x3072."""
    for dt, val in ((torch.float16, 0.37), (torch.float32, 0.6180339)):
        cfa = torch.full((3072, 4096), val, dtype=dt, device=dev)
        rgb = ti.bayer.bayer_to_rgb(cfa)
        assert rgb.shape == (3072, 4096, 3)
        if dt == torch.float16:     # every partial sum k*v is exact in fp32
            assert bool((rgb == cfa[0, 0]).all())
        else:
            assert bool(((rgb - cfa[0, 0]).abs() <= 5e-7 * val).all())


def test_demosaic_torch_container(ti, rng, dev):
    cfa = torch.from_numpy(random_cfa(rng, 32, 64, "f16")).to(dev)
    out = ti.bayer.bayer_to_rgb(cfa)
    assert isinstance(out, torch.Tensor) and out.device == cfa.device and out.dtype == torch.float16
    assert_exact(out.cpu().numpy(), O.bayer_to_rgb(cfa.cpu().numpy()))
    cpu = torch.from_numpy(random_cfa(rng, 8, 8, "f32"))
    assert ti.bayer.bayer_to_rgb(cpu).device.type == "cpu"


def test_demosaic_errors(ti, rng):
    with pytest.raises(AssertionError):
        ti.bayer.bayer_to_rgb(random_cfa(rng, 5, 8, "f32"))
    with pytest.raises(AssertionError):
        ti.bayer.bayer_to_rgb(rng.random((4, 4, 3), dtype=np.float32))
    with pytest.raises(KeyError):
        ti.bayer.bayer_to_rgb(np.zeros((4, 4), np.float64))


# ---------------------------------------------------------------------------------------------
# interpolate.py
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", ["f16", "f32", "u8"])
@pytest.mark.parametrize("scale", [0.46875, 0.8, 1.0, 1.7, (0.3515625, 0.46875)])
def test_resize_bilinear_bit_exact(ti, rng, dtype, scale):
    src = rng.random((61, 83, 3), dtype=np.float32)
    src = (src * 255).astype(np.uint8) if dtype == "u8" else src.astype(O.NP_DTYPE[dtype])
    s = scale if isinstance(scale, tuple) else (scale, scale)
    size = (int(83 * s[1]), int(61 * s[0]))
    assert_exact(ti.interpolate.resize_bilinear(src, size, scale), O.resize_bilinear(src, size, scale),
                 f"resize {dtype} {scale}")


def test_resize_helpers_and_quirk(ti, rng):
    src = rng.random((48, 64, 3), dtype=np.float32)
    assert_exact(ti.interpolate.resize_width(src, 40), O.resize_bilinear(src, (40, int(48 * (40 / 64))), 40 / 64))
    assert_exact(ti.interpolate.scale_bilinear(src, 0.8), O.resize_bilinear(src, (51, 38), 0.8))
    # scale=None: the reference divides (w, h) by (H, W) -- axes crossed (interpolate.py:132-133)
    assert_exact(ti.interpolate.resize_bilinear(src, (32, 24)), O.resize_bilinear(src, (32, 24), None))
    out = ti.interpolate.resize_bilinear(src.astype(np.float16), (32, 24), 0.5, dtype=ti.types.f32)
    assert_exact(out, O.resize_bilinear(src.astype(np.float16), (32, 24), 0.5, "f32"))


@pytest.mark.parametrize("name", O.TRANSFORMS)
def test_transform(ti, rng, name):
    t = ti.ImageTransform(name)
    shape = (9, 9, 3) if name == "transverse" else (7, 12, 3)
    for dt in ("u8", "f16", "f32"):
        src = (rng.random(shape) * 255).astype(O.NP_DTYPE[dt])
        assert_exact(ti.interpolate.transform(src, t), O.transform(src, name), f"transform {name} {dt}")


def test_transform_group_identities(ti, rng):
    T = ti.ImageTransform
    x = (rng.random((6, 10, 3)) * 255).astype(np.uint8)
    tr = ti.interpolate.transform
    assert np.array_equal(tr(tr(x, T.rotate_90), T.rotate_270), x)
    assert np.array_equal(tr(tr(x, T.rotate_180), T.rotate_180), x)
    assert np.array_equal(tr(tr(x, T.transpose), T.transpose), x)
    assert np.array_equal(tr(tr(x, T.flip_horiz), T.flip_vert), tr(x, T.rotate_180))
    with pytest.raises(AssertionError):
        tr(x, T.transverse)


# ---------------------------------------------------------------------------------------------
# tonemap.py (stateless)
# ---------------------------------------------------------------------------------------------
def _rgb_image(rng, H, W, dtype):
    packed = natural_packed12(rng, H, W)
    return O.bayer_to_rgb(O.decode12(packed, dtype, scaled=True))


@pytest.mark.parametrize("din", ["f16", "f32", "u8"])
@pytest.mark.parametrize("dout", ["u8", "f16", "f32"])
@pytest.mark.parametrize("gamma", [1.0, 0.6])
def test_tonemap_linear(ti, rng, din, dout, gamma):
    img = _rgb_image(rng, 50, 70, "f32")
    img = (img * 255).astype(np.uint8) if din == "u8" else img.astype(O.NP_DTYPE[din])
    got = ti.tonemap.tonemap_linear(img, gamma=gamma, dtype=tok(ti, dout))
    assert_close(got, O.tonemap_linear(img, gamma, dout), f"tonemap_linear {din}->{dout}")


@pytest.mark.parametrize("din,dout", [("f16", "u8"), ("f16", "f16"), ("f32", "f32"), ("f32", "u8"), ("u8", "u8")])
@pytest.mark.parametrize("params", [dict(), dict(gamma=0.6), dict(gamma=2.2, intensity=0.5, light_adapt=0.8, color_adapt=0.3)])
def test_tonemap_reinhard(ti, rng, din, dout, params):
    img = _rgb_image(rng, 66, 94, "f32")
    img = (img * 255).astype(np.uint8) if din == "u8" else img.astype(O.NP_DTYPE[din])
    got = ti.tonemap.tonemap_reinhard(img, dtype=tok(ti, dout), **params)
    assert_close(got, O.tonemap_reinhard(img, dtype=dout, **params), f"tonemap_reinhard {din}->{dout} {params}")


@pytest.mark.parametrize("shape", [(480, 640), (302, 518), (1080, 1920)])
def test_tonemap_reinhard_multi_block(ti, rng, shape):
    """Sizes with many blocks per pass (and an odd pixel count): every pass folds the previous pass's per-block
    partials in its prologue, so the block count of the producer and the grid of the consumer both vary."""
    img = _rgb_image(rng, shape[0], shape[1], "f32").astype(np.float16)
    for dout, params in (("f16", dict()), ("u8", dict(gamma=0.6, intensity=1.3, light_adapt=0.9, color_adapt=0.2))):
        got = ti.tonemap.tonemap_reinhard(img, dtype=tok(ti, dout), **params)
        assert_close(got, O.tonemap_reinhard(img, dtype=dout, **params), f"tonemap_reinhard {shape} -> {dout}")
    got = ti.tonemap.tonemap_linear(img, gamma=0.8, dtype=tok(ti, "u8"))
    assert_close(got, O.tonemap_linear(img, 0.8, "u8"), f"tonemap_linear {shape}")


def test_tonemap_reinhard_black_pixels(ti, rng):
    """All-black pixels give 0 * inf = NaN inside the reference formula (light_adapt = 1); the
    library defines NaN -> 0 at the output cast and ignores NaN in the reductions."""
    img = _rgb_image(rng, 32, 48, "f32")
    img[3:9, 5:20] = 0.0
    for dout in ("u8", "f16"):
        got = ti.tonemap.tonemap_reinhard(img, dtype=tok(ti, dout))
        ref = O.tonemap_reinhard(img, dtype=dout)
        assert_close(got, ref, f"black {dout}")
        assert np.all(got[3:9, 5:20] == 0)


# ---------------------------------------------------------------------------------------------
# fused config-2 pipeline (test/pipeline.py:26-32 of the reference)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("p,ids", [(0, False), (1, False), (2, True), (3, False)])
@pytest.mark.parametrize("work,dout", [("f16", "f16"), ("f16", "u8"), ("f32", "f32")])
@pytest.mark.parametrize("shape", [(64, 256), (70, 200), (34, 136), (40, 128)])
def test_pipeline12_reinhard(ti, rng, dev, p, ids, work, dout, shape):
    from taichi_image_amd.pipeline import pipeline12_reinhard
    packed = natural_packed12(rng, *shape, pattern=p, ids_format=ids)
    got = pipeline12_reinhard(torch.from_numpy(packed).to(dev), pat(ti, p), ids, work_dtype=tok(ti, work),
                              dtype=tok(ti, dout), whole_frame=False).cpu().numpy()
    ref = O.pipeline12_reinhard(packed, p, ids, None, work, dout)
    assert_close(got, ref, f"pipeline12 {shape} p{p} {work}->{dout}")


def test_pipeline12_without_work_image_recomputes(ti, rng, dev):
    """The C entry point without scratch for the work-dtype image (work_image_dev == NULL) and an output dtype
    other than the work dtype: every pass re-derives the demosaic from the packed frame (the minimal-traffic
    variant) - same results as the oracle."""
    from taichi_image_amd import _native
    H, W = 70, 200
    packed = natural_packed12(rng, H, W)
    pk = torch.from_numpy(packed).to(dev)
    out = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
    ws = _native.workspace(H, W, dev)
    _native.check(_native.lib().mi_isp_pipeline12_reinhard(
        pk.data_ptr(), out.data_ptr(), None, H, W, 0, 0, None, ti.types.f16.code, ti.types.u8.code, 0.8, 1.0, 1.0, 0.0,
        ws.data_ptr(), _native.stream_ptr(dev)))
    assert_close(out.cpu().numpy(), O.pipeline12_reinhard(packed, 0, False, None, "f16", "u8", gamma=0.8),
                 "pipeline12 recompute variant")


def test_pipeline12_params_and_ccm(ti, rng, dev):
    from taichi_image_amd.pipeline import pipeline12_reinhard
    packed = natural_packed12(rng, 96, 160)
    ccm = O.isp_color_matrix(True, O.DEFAULT_WB, O.DEFAULT_CC)
    kw = dict(gamma=0.6, intensity=1.5, light_adapt=0.7, color_adapt=0.4)
    got = pipeline12_reinhard(torch.from_numpy(packed).to(dev), correct_colors=ccm, **kw, whole_frame=False).cpu().numpy()
    ref = O.pipeline12_reinhard(packed, correct_colors=ccm, **kw)
    assert_close(got, ref, "pipeline12 ccm+params")


def test_specialised_kernels_random_shapes(ti, dev):
    """The compile-time specialisations (packed / plain CFA sources, ragged right and bottom tiles, IDS layout,
    colour matrix) against the NumPy oracle on random even heights and widths that are multiples of 8, plus the
    neighbouring widths that fall back to the generic kernel."""
    from taichi_image_amd.pipeline import pipeline12_reinhard
    r = np.random.default_rng(77)
    ccm = O.isp_color_matrix(True, O.DEFAULT_WB, O.DEFAULT_CC)
    for trial in range(24):
        H = int(r.integers(1, 50)) * 2
        W = int(r.integers(1, 40)) * 8 + (2 if trial % 6 == 5 else 0)          # every 6th: not a multiple of 8
        p = int(r.integers(0, 4)); ids = bool(r.integers(0, 2)); use_ccm = trial % 3 == 0
        packed = natural_packed12(r, H, W, pattern=p, ids_format=ids)
        cc = ccm if use_ccm else None
        got = pipeline12_reinhard(torch.from_numpy(packed).to(dev), pat(ti, p), ids, correct_colors=cc, whole_frame=False).cpu().numpy()
        assert_close(got, O.pipeline12_reinhard(packed, p, ids, cc), f"pipeline12 {H}x{W} p{p} ids={ids} ccm={use_ccm}")
        # demosaic alone: bit-exact, f16 CFA and u16 CFA
        cfa = random_cfa(r, H, W, "f16")
        assert_exact(ti.bayer.bayer_to_rgb(cfa, pat(ti, p)), O.bayer_to_rgb(cfa, p), f"demosaic f16 {H}x{W} p{p}")
        cfa16 = random_cfa(r, H, W, "u16")
        assert_exact(ti.bayer.bayer_to_rgb(cfa16, pat(ti, p)), O.bayer_to_rgb(cfa16, p), f"demosaic u16 {H}x{W} p{p}")


def test_pipeline12_matches_unfused_chain_4k(ti, dev):
    """Full BASELINE size: the fused four-pass pipeline against the unfused GPU chain
    decode12 -> bayer_to_rgb -> tonemap_reinhard (each parity-tested against the oracle above),
    and the demosaic stage against the oracle at full size."""
    from taichi_image_amd.pipeline import pipeline12_reinhard
    from taichi_image_amd.synthetic import synthetic_packed12
    packed = synthetic_packed12(0)
    pk = torch.from_numpy(packed).to(dev)
    fused = pipeline12_reinhard(pk, whole_frame=False)
    cfa = ti.packed.decode12(pk, ti.types.f16, scaled=True)
    rgb = ti.bayer.bayer_to_rgb(cfa)
    unfused = ti.tonemap.tonemap_reinhard(rgb, dtype=ti.types.f16)
    assert fused.shape == (3072, 4096, 3) and fused.dtype == torch.float16
    assert_close(fused.cpu().numpy(), unfused.cpu().numpy(), "fused vs unfused 4K")
    ref_cfa = O.decode12(packed, "f16", scaled=True)
    assert_exact(cfa.cpu().numpy(), ref_cfa, "decode12 4K")
    assert_exact(rgb.cpu().numpy(), O.bayer_to_rgb(ref_cfa), "demosaic 4K")
    # range property of the stateless Reinhard output: normalised to exactly [0, 1]
    assert float(fused.float().min()) == 0.0 and float(fused.float().max()) == 1.0


def test_pipeline12_50mp_matches_unfused_chain(ti, dev):
    """8192 x 6144: more tiles than the 4096-entry floor of the partial rows (the workspace layout scales with
    the tile count) - fused pipeline against the unfused GPU chain, u8 output (recompute variant) as well."""
    from taichi_image_amd.pipeline import pipeline12_reinhard
    from taichi_image_amd.synthetic import synthetic_packed12
    small = synthetic_packed12(2, 1536, 2048)
    packed = np.tile(small, (4, 4))                        # periodic 50 MP frame, cheap to build
    pk = torch.from_numpy(packed).to(dev)
    fused = pipeline12_reinhard(pk, whole_frame=False)
    rgb = ti.bayer.bayer_to_rgb(ti.packed.decode12(pk, ti.types.f16, scaled=True))
    unfused = ti.tonemap.tonemap_reinhard(rgb, dtype=ti.types.f16)
    assert fused.shape == (6144, 8192, 3)
    assert torch.equal(fused, unfused)
    fused8 = pipeline12_reinhard(pk, dtype=ti.types.u8, gamma=0.6, whole_frame=False)
    unfused8 = ti.tonemap.tonemap_reinhard(rgb, gamma=0.6, dtype=ti.types.u8)
    diff = (fused8.int() - unfused8.int()).abs()
    assert int(diff.max()) <= 1 and float((diff > 0).float().mean()) < 1e-3


def test_pipeline12_4k_against_c_oracle(ti, dev):
    """Full BASELINE size against the CPU oracle (the C/OpenMP restatement, ~1 s per frame)."""
    from oracle import c_oracle
    if not c_oracle.available():
        pytest.skip("oracle/liborc_isp.so not built")
    from taichi_image_amd.pipeline import pipeline12_reinhard
    from taichi_image_amd.synthetic import synthetic_packed12
    packed = synthetic_packed12(3)
    got = pipeline12_reinhard(torch.from_numpy(packed).to(dev), whole_frame=False).cpu().numpy()
    assert_close(got, c_oracle.pipeline12_reinhard(packed), "pipeline12 4K vs C oracle")
    got8 = pipeline12_reinhard(torch.from_numpy(packed).to(dev), dtype=ti.types.u8, gamma=0.6, whole_frame=False).cpu().numpy()
    assert_close(got8, c_oracle.pipeline12_reinhard(packed, out="u8", gamma=0.6), "pipeline12 4K u8 vs C oracle")


def test_pipeline12_5mp_sensor_against_c_oracle(ti, dev):
    """A 2448 x 2048 sensor: the width is a multiple of 8 but not of the 128-pixel tile, the height not of 32 -
    the specialised kernels with ragged right / bottom tiles, against the C oracle."""
    from oracle import c_oracle
    if not c_oracle.available():
        pytest.skip("oracle/liborc_isp.so not built")
    from taichi_image_amd.pipeline import pipeline12_reinhard
    from taichi_image_amd.synthetic import synthetic_packed12
    packed = synthetic_packed12(5, 2050, 2448)
    got = pipeline12_reinhard(torch.from_numpy(packed).to(dev), whole_frame=False).cpu().numpy()
    assert_close(got, c_oracle.pipeline12_reinhard(packed), "pipeline12 2448x2050 vs C oracle")


def test_batch_pipeline_equals_single(ti, rng, dev):
    from taichi_image_amd.pipeline import BatchPipeline, pipeline12_reinhard
    H, W = 64, 128
    frames = [torch.from_numpy(natural_packed12(rng, H, W)).to(dev) for _ in range(5)]
    bp = BatchPipeline(5, H, W, dev, n_streams=3, whole_frame=False)
    outs = [o.clone() for o in bp(frames)]
    torch.cuda.synchronize()
    for f, o in zip(frames, outs):
        assert torch.equal(o, pipeline12_reinhard(f, whole_frame=False))


def test_batch_pipeline_graph_replay(ti, rng, dev):
    """use_graph: the captured step replays with new contents in the same input buffers, recaptures
    when the buffers change, and an eager step in between gives the same results."""
    from taichi_image_amd.pipeline import BatchPipeline, pipeline12_reinhard
    H, W = 64, 128
    host = [natural_packed12(rng, H, W) for _ in range(8)]
    bufs = [torch.from_numpy(host[i]).to(dev) for i in range(4)]
    bp = BatchPipeline(4, H, W, dev, n_streams=2, use_graph=True, whole_frame=False)
    for round_ in range(3):
        for i in range(4):
            bufs[i].copy_(torch.from_numpy(host[(i + round_) % 8]))       # same addresses, new frames
        outs = [o.clone() for o in bp(bufs, eager=(round_ == 1))]
        torch.cuda.synchronize()
        for i, o in enumerate(outs):
            assert torch.equal(o, pipeline12_reinhard(torch.from_numpy(host[(i + round_) % 8]).to(dev), whole_frame=False))
    other = [torch.from_numpy(host[4 + i]).to(dev) for i in range(4)]         # different buffers: recapture
    outs = [o.clone() for o in bp(other)]
    torch.cuda.synchronize()
    for i, o in enumerate(outs):
        assert torch.equal(o, pipeline12_reinhard(other[i], whole_frame=False))


# ---------------------------------------------------------------------------------------------
# camera_isp.py (stateful ISP)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cam,work", [("Camera16", "f16"), ("Camera32", "f32")])
@pytest.mark.parametrize("kw", [dict(), dict(resize_width=96), dict(scale=0.5), dict(correct_colors=True, resize_width=80)])
def test_isp_load_packed12_bit_exact(ti, rng, dev, cam, work, kw):
    packed = natural_packed12(rng, 98, 168)
    isp = getattr(ti, cam)(ti.BayerPattern.RGGB, device=dev, **kw)
    got = isp.load_packed12(torch.from_numpy(packed).to(dev))
    ccm = O.isp_color_matrix(kw.get("correct_colors", False), O.DEFAULT_WB, O.DEFAULT_CC)
    ref = O.isp_load_packed12(packed, work, correct_colors=ccm, resize_width=kw.get("resize_width", 0),
                              scale=kw.get("scale"))
    assert_exact(got.cpu().numpy(), ref, f"{cam}.load_packed12 {kw}")


@pytest.mark.parametrize("p", [0, 1, 2, 3])
@pytest.mark.parametrize("shape,kw", [((98, 168), dict(scale=1.3)), ((98, 168), dict(scale=0.3)),      # upscale; unfused fallback
                                      ((70, 170), dict(resize_width=90)), ((34, 136), dict(scale=0.46875)),
                                      ((130, 264), dict(resize_width=124, correct_colors=True))])
def test_isp_load_packed12_fused_resize(ti, rng, dev, p, shape, kw):
    """unpack -> demosaic -> bilinear in one kernel (csrc/isp_resize_tile.h): every pattern, widths
    that are not a multiple of 8 (byte-path fill), borders, colour matrix, up- and down-scaling."""
    packed = natural_packed12(rng, *shape, pattern=p)
    isp = ti.Camera16(pat(ti, p), device=dev, **kw)
    got = isp.load_packed12(torch.from_numpy(packed).to(dev))
    ccm = O.isp_color_matrix(kw.get("correct_colors", False), O.DEFAULT_WB, O.DEFAULT_CC)
    ref = O.isp_load_packed12(packed, "f16", p, correct_colors=ccm, resize_width=kw.get("resize_width", 0),
                              scale=kw.get("scale"))
    assert_exact(got.cpu().numpy(), ref, f"fused resize {shape} p{p} {kw}")


def test_isp_load_packed12_random_scales(ti, dev):
    """Camera16 / Camera32 load_packed12 with random sizes, patterns and scales (fused kernel where the scale
    fits, the unfused chain elsewhere), bit-exact against the oracle."""
    r = np.random.default_rng(99)
    for trial in range(12):
        H = int(r.integers(8, 60)) * 2
        W = int(r.integers(4, 30)) * 8 + (2 * int(r.integers(0, 4)) if trial % 4 == 3 else 0)
        p = int(r.integers(0, 4))
        scale = float(np.float32(r.uniform(0.25, 1.6)))
        work, cam = (("f16", "Camera16") if trial % 3 else ("f32", "Camera32"))
        packed = natural_packed12(r, H, W, pattern=p)
        isp = getattr(ti, cam)(pat(ti, p), device=dev, scale=scale)
        got = isp.load_packed12(torch.from_numpy(packed).to(dev))
        ref = O.isp_load_packed12(packed, work, p, scale=scale)
        assert_exact(got.cpu().numpy(), ref, f"{cam} load_packed12 {H}x{W} p{p} scale={scale}")


def test_isp_load_packed12_config3_full_size(ti, dev):
    """BASELINE config 3: 4096x3072 packed-12 -> Camera16(resize_width=1920) -> f16 (1440, 1920, 3)."""
    from oracle import c_oracle
    from taichi_image_amd.synthetic import synthetic_packed12
    packed = synthetic_packed12(1)
    isp = ti.Camera16(ti.BayerPattern.RGGB, resize_width=1920, device=dev)
    got = isp.load_packed12(torch.from_numpy(packed).to(dev)).cpu().numpy()
    assert got.shape == (1440, 1920, 3) and got.dtype == np.float16
    if c_oracle.available():      # full-size demosaic from the C oracle (fast), resize from the NumPy oracle
        cfa = c_oracle.decode12_scaled(packed, work="f16").reshape(3072, 4096)
        rgb = c_oracle.demosaic(cfa, 0, round_f16=True).astype(np.float16)
    else:
        rgb = O.bayer_to_rgb(O.decode12(packed, "f16", scaled=True))
    assert_exact(got, O.resize_bilinear(rgb, (1920, 1440), 1920 / 4096), "config 3 load_packed12")


def test_isp_other_loaders(ti, rng, dev):
    isp = ti.Camera16(ti.BayerPattern.GBRG, device=dev, resize_width=40)
    raw16 = rng.integers(0, 65536, (48, 64)).astype(np.uint16)
    enc = raw16.view(np.uint8).reshape(48, 128)
    ref = O.isp_load_packed16(enc, "f16", O.GBRG, resize_width=40)
    assert_exact(isp.load_packed16(torch.from_numpy(enc).to(dev)).cpu().numpy(), ref, "load_packed16")
    sz, s = O.isp_output_size(48, 64, 40)
    for name, fn in (("load_16u", O.load_16u), ("load_16f", O.load_16f)):
        src = raw16 if name == "load_16u" else rng.integers(0, 2, (48, 64)).astype(np.uint16)
        want = O.resize_bilinear(O.bayer_to_rgb(fn(src, "f16"), O.GBRG), sz, s)
        assert_exact(getattr(isp, name)(torch.from_numpy(src).to(dev)).cpu().numpy(), want, name)
    f = rng.random((48, 64), dtype=np.float32)
    want = O.resize_bilinear(O.bayer_to_rgb(O.load_32f(f, "f16"), O.GBRG), sz, s)
    assert_exact(isp.load_32f(torch.from_numpy(f).to(dev)).cpu().numpy(), want, "load_32f")
    ids = natural_packed12(rng, 48, 64, O.GBRG, ids_format=True)
    assert_exact(isp.load_packed12(torch.from_numpy(ids).to(dev), ids_format=True).cpu().numpy(),
                 O.isp_load_packed12(ids, "f16", O.GBRG, True, resize_width=40), "load_packed12 ids")


@pytest.mark.parametrize("cam,work", [("Camera16", "f16"), ("Camera32", "f32")])
def test_isp_tonemap_reinhard_sequence(ti, rng, dev, cam, work):
    """test/camera_isp.py:29-39 call sequence over three steps so the moving average is exercised."""
    isp = getattr(ti, cam)(ti.BayerPattern.RGGB, moving_alpha=0.3, resize_width=64, device=dev)
    st = O.IspState(0.3)
    for step in range(3):
        packs = [natural_packed12(rng, 80, 128, dark=0.05 * step) for _ in range(3)]
        imgs = [isp.load_packed12(torch.from_numpy(p).to(dev)) for p in packs]
        refs = [O.isp_load_packed12(p, work, resize_width=64) for p in packs]
        outs = isp.tonemap_reinhard(imgs, gamma=0.6)
        m = st.update_metering(refs)
        assert_close(isp.metrics.cpu().numpy(), m, f"metrics step {step}", rel=2e-5)
        for k, (o, im, r) in enumerate(zip(outs, imgs, refs)):
            ref_u8, ref_after = O.reinhard_isp(r, m, gamma=0.6)
            assert_close(o.cpu().numpy(), ref_u8, f"u8 step {step} img {k}")
            assert_close(im.cpu().numpy(), ref_after, f"in-place p step {step} img {k}")


def test_isp_tonemap_linear_and_transforms(ti, rng, dev):
    packs = [natural_packed12(rng, 64, 64) for _ in range(2)]
    for name in O.TRANSFORMS:
        isp = ti.Camera16(ti.BayerPattern.RGGB, moving_alpha=1.0, device=dev, transform=ti.ImageTransform(name))
        imgs = [isp.load_packed12(torch.from_numpy(p).to(dev)) for p in packs]
        refs = [O.isp_load_packed12(p, "f16") for p in packs]
        m = O.IspState(1.0).update_metering(refs)
        lin = isp.tonemap_linear([im.clone() for im in imgs], gamma=0.8)
        for o, r in zip(lin, refs):
            assert_close(o.cpu().numpy(), O.transform(O.linear_isp(r, m, 0.8), name), f"linear {name}")
        isp.metrics = None
        rh = isp.tonemap_reinhard(imgs, gamma=1.0, intensity=0.7, light_adapt=0.9, color_adapt=0.2)
        for o, r in zip(rh, refs):
            ref_u8, _ = O.reinhard_isp(r, m, 1.0, 0.7, 0.9, 0.2)
            assert_close(o.cpu().numpy(), O.transform(ref_u8, name), f"reinhard {name}")


def test_isp_constructor_and_set(ti, dev):
    with pytest.raises(AssertionError):
        ti.Camera16(ti.BayerPattern.RGGB, scale=0.5, resize_width=100, device=dev)
    with pytest.raises(TypeError):
        ti.Camera16("RGGB", device=dev)
    isp = ti.Camera16(ti.BayerPattern.RGGB, resize_width=100, device=dev)
    isp.set(scale=0.5)
    assert isp.scale == 0.5 and isp.resize_width == 0
    isp.set(resize_width=50)
    assert isp.scale is None and isp.resize_width == 50
    with pytest.raises(TypeError):
        isp.tonemap_reinhard([], gamma=1)
    assert ti.Camera16.__qualname__ == "Camera16" and callable(ti.Camera16.reinhard_kernel)


@pytest.mark.parametrize("cam,work", [("Camera16", "f16"), ("Camera32", "f32")])
def test_isp_reference_quirks_drop_the_pattern(ti, rng, dev, cam, work):
    """ISP._process_image of the reference calls bayer_to_rgb without its pattern (camera_isp.py:371-373): every camera
    is demosaiced as RGGB.  reference_quirks=True reproduces that (strict comparison with the reference's outputs), the
    default honours bayer_pattern; for an RGGB camera the two agree."""
    H, W = 64, 96
    packed = natural_packed12(rng, H, W, pattern=O.BGGR)
    frame = torch.from_numpy(packed).to(dev)
    honest = getattr(ti, cam)(ti.BayerPattern.BGGR, device=dev).load_packed12(frame)
    quirky = getattr(ti, cam)(ti.BayerPattern.BGGR, device=dev, reference_quirks=True).load_packed12(frame)
    assert_exact(honest.cpu().numpy(), O.isp_load_packed12(packed, work, pattern=O.BGGR), "pattern honoured")
    assert_exact(quirky.cpu().numpy(), O.isp_load_packed12(packed, work, pattern=O.RGGB), "pattern dropped, as the reference does")
    assert not torch.equal(honest, quirky)
    cfa16 = rng.integers(0, 65536, (H, W)).astype(np.uint16)
    q16 = getattr(ti, cam)(ti.BayerPattern.GRBG, device=dev, reference_quirks=True).load_16u(torch.from_numpy(cfa16))
    r16 = getattr(ti, cam)(ti.BayerPattern.RGGB, device=dev).load_16u(torch.from_numpy(cfa16))
    assert torch.equal(q16, r16)


# ---- color/yuv_420.py (the step after the path) --------------------------------------------------
@pytest.mark.parametrize("in_dt,out_dt", [("u8", None), ("f16", "u8"), ("f32", "f16"), ("u16", "u8"), ("f32", None)])
@pytest.mark.parametrize("shape", [(2, 2), (6, 10), (64, 96)])
def test_rgb_yuv420_matches_oracle(rng, dev, in_dt, out_dt, shape):
    from taichi_image_amd import color
    h, w = shape
    if in_dt in ("u8", "u16"):
        img = rng.integers(0, int(O.SCALE[in_dt]) + 1, (h, w, 3)).astype(O.NP_DTYPE[in_dt])
    else:
        # some values outside [0, 1]; a negative float -> unsigned cast is undefined in the reference,
        # oracle and library both saturate it to 0
        img = (rng.random((h, w, 3)) * 1.2 - 0.05).astype(O.NP_DTYPE[in_dt])
    want = O.rgb_yuv420(img, out_dt)
    got = color.rgb_yuv420_image(img, dtype=out_dt)
    assert isinstance(got, np.ndarray)
    assert_close(got, want, f"rgb_yuv420 {in_dt}->{out_dt} {shape}")
    t = torch.from_numpy(img).to(dev)
    got_t = color.rgb_yuv420_image(t, dtype=out_dt)
    assert got_t.device == t.device
    assert_exact(got_t.cpu().numpy(), got, "torch container gives the same result")


@pytest.mark.parametrize("in_dt,out_dt", [("u8", None), ("u8", "f32"), ("f16", None), ("f32", "u8")])
def test_yuv420_rgb_matches_oracle(rng, dev, in_dt, out_dt):
    from taichi_image_amd import color
    h, w = 12, 16
    rgb = rng.integers(0, 256, (h, w, 3)).astype(np.uint8)
    yuv8 = O.rgb_yuv420(rgb)
    yuv = yuv8 if in_dt == "u8" else (yuv8.astype(np.float32) / 255).astype(O.NP_DTYPE[in_dt])
    want = O.yuv420_rgb(yuv, out_dt)
    got = color.yuv420_rgb_image(yuv, dtype=out_dt)
    assert_close(got, want, f"yuv420_rgb {in_dt}->{out_dt}")
    y, uv, (ww, hh) = color.split_yuv_420(got if got.ndim == 2 else yuv)
    assert (ww, hh) == (w, h) and uv.shape == (2, h // 2, w // 2)


def test_yuv420_4k_tonemap_output_round_trip(dev):
    """Full-size property: u8 frame -> yuv420 -> rgb stays within the chroma-subsampling error of a
    smooth scene, and the Y plane equals the per-pixel luma of the oracle on a crop."""
    from taichi_image_amd import color, synthetic
    scene = (np.clip(synthetic.synthetic_scene(3), 0, 1) * 255).astype(np.uint8)
    t = torch.from_numpy(scene).to(dev)
    yuv = color.rgb_yuv420_image(t)
    assert yuv.shape == (scene.shape[0] * 3 // 2, scene.shape[1]) and yuv.dtype == torch.uint8
    crop = scene[:64, :128]
    assert_close(yuv[:64, :128].cpu().numpy(), O.rgb_yuv420(crop)[:64], "luma plane crop")
    back = color.yuv420_rgb_image(yuv).cpu().numpy().astype(np.int32)
    assert np.abs(back - scene.astype(np.int32)).mean() < 8.0


def test_yuv420_asserts():
    from taichi_image_amd import color
    with pytest.raises(AssertionError):
        color.rgb_yuv420_image(np.zeros((3, 4, 3), np.uint8))
    with pytest.raises(AssertionError):
        color.rgb_yuv420_image(np.zeros((4, 4), np.uint8))


# ---- ingest path (scripts/tonemap_scan.py:64-87) --------------------------------------------------
def test_upload_ring_pipeline_matches_direct(dev):
    """Frames uploaded through the pinned ring while earlier frames are still being processed give
    the results of frames that were resident from the start."""
    from taichi_image_amd import ingest, synthetic
    from taichi_image_amd.pipeline import pipeline12_reinhard
    H, W = 64, 256
    frames = [synthetic.synthetic_packed12(k, H, W) for k in range(5)]
    ring = ingest.UploadRing(2, H * W * 3 // 2, dev)
    outs = []
    for f in frames:                                   # more frames than slots: slots are reused
        slot, dev_bytes = ring.upload(f)
        outs.append(pipeline12_reinhard(dev_bytes.view(H, W * 3 // 2), whole_frame=False))
        slot.release()
    torch.cuda.synchronize()
    for f, o in zip(frames, outs):
        want = pipeline12_reinhard(torch.from_numpy(f).to(dev), whole_frame=False)
        assert_exact(o.cpu().numpy(), want.cpu().numpy(), "ring-fed frame")


def test_load_raw_bytes_and_iter(dev, tmp_path):
    from pathlib import Path
    from taichi_image_amd import ingest
    rng = np.random.default_rng(5)
    folders = [tmp_path / "cam0", tmp_path / "cam1"]
    names = ["a.raw", "b.raw", "c.raw"]
    blobs = {}
    for fo in folders:
        fo.mkdir()
        for n in names:
            blobs[(fo, n)] = rng.integers(0, 256, 6144, dtype=np.uint8)
            (fo / n).write_bytes(blobs[(fo, n)].tobytes())
    t = ingest.load_raw_bytes(folders[0] / "a.raw", device=dev)
    assert t.device == dev and t.dtype == torch.uint8
    assert_exact(t.cpu().numpy(), blobs[(folders[0], "a.raw")], "raw bytes")
    from functools import partial
    seen = []
    for name, group in ingest.load_images_iter(partial(ingest.load_raw_bytes, device=dev), folders, names):
        seen.append(name)
        for fo in folders:
            assert_exact(group[fo].cpu().numpy(), blobs[(fo, name)], f"{fo.name}/{name}")
    assert seen == names


def test_isp_sharded_metering_path_equals_fused(ti, rng, dev):
    """The multi-GPU variant of update_metering (mi_isp_metering_bounds -> exchange -> mi_isp_metering_sums
    -> exchange, taichi_image_amd/distributed.py) on one rank equals the single-call path: same
    statistics, same outputs over three consecutive calls (the moving average is exercised)."""
    H, W = 96, 160
    frames = [torch.from_numpy(natural_packed12(rng, H, W)).to(dev) for _ in range(3)]
    a = ti.Camera16(ti.BayerPattern.RGGB, moving_alpha=0.3, device=dev)
    b = ti.Camera16(ti.BayerPattern.RGGB, moving_alpha=0.3, device=dev, process_group=object())   # one "rank"
    for step in range(3):
        fa = [a.load_packed12(f) for f in frames]
        fb = [b.load_packed12(f) for f in frames]
        oa = a.tonemap_reinhard(fa, gamma=0.6)
        ob = b.tonemap_reinhard(fb, gamma=0.6)
        assert_close(b.metrics.cpu().numpy(), a.metrics.cpu().numpy(), f"metering state, call {step}", rel=1e-5)
        for x, y in zip(oa, ob):
            assert_close(y.cpu().numpy(), x.cpu().numpy(), f"u8 output, call {step}")


@pytest.mark.parametrize("shape,n", [((96, 160), 3), ((8, 16), 1), ((200, 520), 6), ((768, 1024), 6)])
def test_isp_metering_in_one_launch_equals_four_launches(ti, rng, dev, shape, n, monkeypatch):
    """update_metering as ONE kernel with a grid barrier inside (mi_isp_metering's default) against the four launches it
    replaces (MI_ISP_METERING_LAUNCHES=4: bounds pass, finalize, statistics pass, finalize) and against the oracle: the
    rolling state over three calls, and the u8 outputs that follow from it."""
    H, W = shape
    frames = [torch.from_numpy(natural_packed12(rng, H, W)).to(dev) for _ in range(n)]
    a = ti.Camera16(ti.BayerPattern.RGGB, moving_alpha=0.3, device=dev)
    b = ti.Camera16(ti.BayerPattern.RGGB, moving_alpha=0.3, device=dev)
    for step in range(3):
        monkeypatch.delenv("MI_ISP_METERING_LAUNCHES", raising=False)
        fa = [a.load_packed12(f) for f in frames]
        oa = a.tonemap_reinhard(fa, gamma=0.6)
        torch.cuda.synchronize()
        monkeypatch.setenv("MI_ISP_METERING_LAUNCHES", "4")
        fb = [b.load_packed12(f) for f in frames]
        ob = b.tonemap_reinhard(fb, gamma=0.6)
        torch.cuda.synchronize()
        assert_close(a.metrics.cpu().numpy(), b.metrics.cpu().numpy(), f"metering state, call {step}", rel=1e-5)
        for x, y in zip(oa, ob):
            assert_close(x.cpu().numpy(), y.cpu().numpy(), f"u8 output, call {step}")
    monkeypatch.delenv("MI_ISP_METERING_LAUNCHES", raising=False)
    from taichi_image_amd import _native
    ws = _native.workspace(H, W, dev)
    off = int(_native.lib().mi_isp_workspace_error_offset(H, W))
    assert int(ws[off:off + 4].view(torch.int32).item()) == 0, "the metering kernel's grid barrier timed out"


def test_isp_metering_kernels_of_several_streams_are_put_in_order(ti, rng, dev):
    """Four camera groups on four streams at once: the one-launch metering kernels (a grid barrier inside each) are
    serialised by the library, so every group's state and outputs equal those of the same group run alone."""
    from taichi_image_amd import _native
    H, W = 768, 1024
    groups = [[torch.from_numpy(natural_packed12(rng, H, W, dark=0.03 * g)).to(dev) for _ in range(6)] for g in range(4)]
    alone = []
    for frames in groups:
        cam = ti.Camera16(ti.BayerPattern.RGGB, moving_alpha=0.3, device=dev)
        for _ in range(2):
            outs = cam.tonemap_reinhard([cam.load_packed12(f) for f in frames], gamma=0.6)
        torch.cuda.synchronize()
        alone.append((cam.metrics.clone(), [o.clone() for o in outs]))
    streams = [torch.cuda.Stream(device=dev) for _ in groups]
    cams = [ti.Camera16(ti.BayerPattern.RGGB, moving_alpha=0.3, device=dev) for _ in groups]
    torch.cuda.synchronize()
    res = [None] * len(groups)
    for _ in range(2):
        for g, (frames, st_) in enumerate(zip(groups, streams)):
            with torch.cuda.stream(st_):
                res[g] = cams[g].tonemap_reinhard([cams[g].load_packed12(f) for f in frames], gamma=0.6)
    torch.cuda.synchronize()
    for g in range(len(groups)):
        assert torch.equal(cams[g].metrics, alone[g][0]), f"group {g}: metering state"
        for x, y in zip(res[g], alone[g][1]):
            assert torch.equal(x, y), f"group {g}: u8 output"
        ws = None
    for st_ in streams:
        with torch.cuda.stream(st_):
            ws = _native.workspace(H, W, dev)
            off = int(_native.lib().mi_isp_workspace_error_offset(H, W))
            assert int(ws[off:off + 4].view(torch.int32).item()) == 0, "a metering kernel's grid barrier timed out"


def test_isp_metering_timeout_leaves_the_state_alone(ti, dev):
    """The one-launch update_metering with a poll budget of one round: some block cannot have seen all its peers.  The
    call must fail as a whole - metrics exactly as they were, fault word and mailbox set, MeteringTimeout at the next call -
    and never blend bounds folded from the records that happened to be there into the rolling state."""
    from oracle import c_oracle
    from taichi_image_amd import _native
    from taichi_image_amd.camera_isp import MeteringTimeout
    L = _native.lib()
    H, W = 768, 1024                                                # 6 images: 12 blocks
    frames = [torch.from_numpy(natural_packed12(np.random.default_rng(500 + k), H, W, dark=0.02 * k)).to(dev) for k in range(6)]
    isp = ti.Camera16(ti.BayerPattern.RGGB, moving_alpha=0.3, device=dev)
    imgs = [isp.load_packed12(f) for f in frames]
    refs = [im.cpu().numpy() for im in imgs]
    st = c_oracle.IspState(0.3)
    isp.update_metering(imgs)
    m1 = st.update_metering(refs)
    before = isp.metrics.clone()
    assert_close(before.cpu().numpy(), m1, "metrics 1", rel=2e-5)
    assert L.mi_isp_metering_faults(1) == 0
    # other images (the state WOULD move), without the load kernels' subsample
    others = [(im * 0.5).contiguous() for im in imgs]
    L.mi_isp_metering_set_poll_limit(1)
    try:
        isp.update_metering(others)
        torch.cuda.synchronize()
    finally:
        L.mi_isp_metering_set_poll_limit(0)
    assert torch.equal(isp.metrics, before), "a timed-out metering call changed the state"
    assert L.mi_isp_metering_faults(0) != 0
    ws = _native.workspace(H, W, dev)
    off = int(L.mi_isp_workspace_error_offset(H, W))
    assert int(ws[off:off + 4].view(torch.int32).item()) != 0
    ws[off:off + 4].zero_()
    with pytest.raises(MeteringTimeout):
        isp.update_metering(imgs)
    isp.update_metering(others)                                     # the mailbox was cleared by the report: back to normal
    assert not torch.equal(isp.metrics, before)
    assert_close(isp.metrics.cpu().numpy(), st.update_metering([o.cpu().numpy() for o in others]), "metrics after the failed call", rel=2e-5)


@pytest.mark.parametrize("cam,shape,n,kw", [("Camera16", (48, 64), 1, dict(gamma=0.6)), ("Camera16", (96, 128), 2, dict()),
                                            ("Camera16", (200, 512), 3, dict(gamma=0.8, intensity=1.3, light_adapt=0.7)),
                                            ("Camera32", (96, 128), 7, dict(gamma=0.6)),
                                            ("Camera16", (768, 1024), 6, dict(gamma=0.6, color_adapt=0.4)),
                                            ("Camera32", (768, 1024), 4, dict(gamma=2.2, color_adapt=0.3, light_adapt=0.5))])
def test_isp_reinhard_in_one_launch_equals_two_passes(ti, dev, cam, shape, n, kw, monkeypatch):
    """mi_isp_reinhard_batch as one persistent launch (p kept on chip between the two passes of camera_isp.py:198-218,
    the images' max_out exchanged through per-XCD words) against the two launches it replaces: in-place p and u8 outputs
    bit for bit - one image, pairs, odd counts (the pipeline's tail), both dtypes, color_adapt != 0 - and against the oracle."""
    from oracle import c_oracle
    H, W = shape
    frames = [torch.from_numpy(natural_packed12(np.random.default_rng(700 + k), H, W, dark=0.03 * k)).to(dev) for k in range(n)]

    def run(two_pass):
        monkeypatch.setenv("MI_ISP_REINHARD_LAUNCHES", "2" if two_pass else "1")
        isp = getattr(ti, cam)(ti.BayerPattern.RGGB, moving_alpha=0.4, device=dev)
        res = []
        for step in range(2):
            imgs = [isp.load_packed12(f) for f in frames]
            before = [im.clone() for im in imgs]
            outs = isp.tonemap_reinhard(imgs, **kw)
            res.append((before, [im.clone() for im in imgs], [o.clone() for o in outs], isp.metrics.clone()))
        torch.cuda.synchronize()
        return res
    one, two = run(False), run(True)
    for step in range(2):
        assert torch.equal(one[step][3], two[step][3])
        for k in range(n):
            bits = torch.int16 if cam == "Camera16" else torch.int32      # (bit patterns: p may hold NaN - black pixels)
            assert torch.equal(one[step][1][k].view(bits), two[step][1][k].view(bits)), f"step {step} image {k}: in-place p differs"
            assert torch.equal(one[step][2][k], two[step][2][k]), f"step {step} image {k}: u8 output differs"
    if c_oracle.available():
        before, after, outs, metrics = one[1]
        ref_u8, ref_after = c_oracle.reinhard_isp(before[n - 1].cpu().numpy(), metrics.cpu().numpy(), **kw)
        assert_close(outs[n - 1].cpu().numpy(), ref_u8, "u8 against the oracle")
        assert_close(after[n - 1].cpu().numpy(), ref_after, "in-place p against the oracle")
    from taichi_image_amd import _native
    assert _native.lib().mi_isp_reinhard_faults(1) == 0


@pytest.mark.parametrize("cam,shape,n,kw", [("Camera16", (200, 512), 3, dict(gamma=0.6)), ("Camera32", (96, 128), 2, dict(gamma=2.2, color_adapt=0.3)),
                                            ("Camera16", (70, 200), 2, dict()), ("Camera16", (768, 1024), 6, dict(gamma=0.8, intensity=1.2, light_adapt=0.6))])
def test_isp_tonemap_reinhard_without_write_back(ti, dev, cam, shape, n, kw):
    """Extension: tonemap_reinhard(write_back=False) gives the u8 outputs and metrics of the default call bit for bit and
    leaves the images (and their metering subsample tags) untouched - the default overwrites them (camera_isp.py:211)."""
    H, W = shape
    frames = [torch.from_numpy(natural_packed12(np.random.default_rng(900 + k), H, W, dark=0.03 * k)).to(dev) for k in range(n)]
    for transform in (ti.interpolate.ImageTransform.none, ti.interpolate.ImageTransform.rotate_90):
        a = getattr(ti, cam)(ti.BayerPattern.RGGB, moving_alpha=0.4, transform=transform, device=dev)
        b = getattr(ti, cam)(ti.BayerPattern.RGGB, moving_alpha=0.4, transform=transform, device=dev)
        for step in range(2):
            ia, ib = [a.load_packed12(f) for f in frames], [b.load_packed12(f) for f in frames]
            before = [im.clone() for im in ib]
            tags = [hasattr(im, "_mi_metering_sub") for im in ib]
            oa = a.tonemap_reinhard(ia, **kw)
            ob = b.tonemap_reinhard(ib, write_back=False, **kw)
            assert torch.equal(a.metrics, b.metrics)
            bits = torch.int16 if cam == "Camera16" else torch.int32
            for k in range(n):
                assert torch.equal(oa[k], ob[k]), f"step {step} image {k}: u8 output differs"
                assert torch.equal(ib[k].view(bits), before[k].view(bits)), "write_back=False changed the image"
                assert hasattr(ib[k], "_mi_metering_sub") == tags[k]
                assert not torch.equal(ia[k].view(bits), before[k].view(bits)), "the default call did not write p back"


def test_isp_reinhard_one_launch_timeout_is_reported(ti, dev, monkeypatch):
    """The fused tonemap's wait for max_out with a budget of one poll: a block that looks before the others have arrived
    gives up - fault word, mailbox, TonemapTimeout at the next call - and nothing hangs; the call after that is clean."""
    from taichi_image_amd import _native
    from taichi_image_amd.camera_isp import TonemapTimeout
    L = _native.lib()
    monkeypatch.setenv("MI_ISP_REINHARD_LAUNCHES", "1")           # (the one-launch form is opt-in: measured slower)
    H, W = 768, 1024
    frames = [torch.from_numpy(natural_packed12(np.random.default_rng(800 + k), H, W)).to(dev) for k in range(4)]
    isp = ti.Camera16(ti.BayerPattern.RGGB, moving_alpha=0.4, device=dev)
    good = isp.tonemap_reinhard([isp.load_packed12(f) for f in frames], gamma=0.6)
    torch.cuda.synchronize()
    assert L.mi_isp_reinhard_faults(1) == 0
    L.mi_isp_reinhard_set_poll_limit(1)
    try:
        # (one image: its wait follows its own pass at once - in the pipelined form a whole pass of the next image lies
        # between, and every block has usually arrived by then)
        isp.tonemap_reinhard([isp.load_packed12(frames[0])], gamma=0.6)
        torch.cuda.synchronize()
    finally:
        L.mi_isp_reinhard_set_poll_limit(0)
    assert L.mi_isp_reinhard_faults(0) != 0
    ws = _native.workspace(H, W, dev)
    off = int(L.mi_isp_workspace_error_offset(H, W))
    assert int(ws[off:off + 4].view(torch.int32).item()) != 0
    ws[off:off + 4].zero_()
    with pytest.raises(TonemapTimeout):
        isp.update_metering([isp.load_packed12(frames[0])])
    isp2 = ti.Camera16(ti.BayerPattern.RGGB, moving_alpha=0.4, device=dev)
    again = isp2.tonemap_reinhard([isp2.load_packed12(f) for f in frames], gamma=0.6)
    assert all(torch.equal(a, b) for a, b in zip(again, good)) and L.mi_isp_reinhard_faults(0) == 0


@pytest.mark.parametrize("shape,n,pattern,kw,cc", [
    ((48, 64), 3, "RGGB", dict(), False), ((36, 520), 2, "GRBG", dict(gamma=0.6), True),
    ((100, 1032), 4, "BGGR", dict(gamma=0.6, color_adapt=0.3, intensity=1.2, light_adapt=0.7), True),
    ((768, 1024), 6, "GBRG", dict(gamma=2.2), False), ((26, 4096), 2, "RGGB", dict(gamma=0.6), False)])
def test_isp_process_packed12_equals_the_two_calls(ti, dev, shape, n, pattern, kw, cc):
    """Extension: ISP.process_packed12 - the reference bench's Processor step (bench/camera_isp.py:23-27) in one call, on the
    camera-group kernel (csrc/isp_mega_cam.h: subsample from the packed frames, metering, ONE persistent launch from packed
    bytes to u8) - gives the u8 outputs, the images the reference leaves behind (p, camera_isp.py:211) and the metering
    state of `tonemap_reinhard([load_packed12(f) ...])` bit for bit, over three groups of a rolling metering; rows that do
    not fill a wave's 12, bands narrower than 512 columns, image borders, all four patterns, the colour matrix."""
    from taichi_image_amd import _native
    L = _native.lib()
    H, W = shape
    pat = getattr(ti.BayerPattern, pattern)
    assert L.mi_isp_camera_group_fits(H, W, pat.value, ti.types.f16.code, 8) == 1
    a = ti.Camera16(pat, moving_alpha=0.3, correct_colors=cc, device=dev)
    b = ti.Camera16(pat, moving_alpha=0.3, correct_colors=cc, device=dev)
    for group in range(3):
        frames = [torch.from_numpy(natural_packed12(np.random.default_rng(1000 + 10 * group + k), H, W, dark=0.03 * k)).to(dev)
                  for k in range(n)]
        keep = group != 1                                 # (the bench's form - nothing kept - in the middle group)
        got = a.process_packed12(frames, keep_images=keep, **kw)
        outs, images = got if keep else (got, None)
        want_images = [b.load_packed12(f) for f in frames]
        want = b.tonemap_reinhard(want_images, **kw)
        torch.cuda.synchronize()
        assert L.mi_isp_camera_group_faults(0) == 0
        assert torch.equal(a.metrics.view(torch.int32), b.metrics.view(torch.int32)), f"group {group}: metering state"
        for k in range(n):
            assert torch.equal(outs[k], want[k]), f"group {group} camera {k}: u8 output"
            if keep:
                assert torch.equal(images[k].view(torch.int16), want_images[k].view(torch.int16)), f"group {group} camera {k}: p"


def test_isp_metrics_are_rebound_not_overwritten(ti, dev):
    """camera_isp.py:172-173,376-385: every update leaves `isp.metrics` bound to a NEW tensor and the previous one as it was
    (the reference clones it first).  Round 4 dropped the clone - the kernel reads the old state and writes the new tensor -
    so the property is tested: through tonemap_reinhard, update_metering and process_packed12."""
    H, W = 96, 512
    fr = [[torch.from_numpy(natural_packed12(np.random.default_rng(1400 + 10 * s + k), H, W, dark=0.05 * s)).to(dev) for k in range(2)]
          for s in range(4)]
    isp = ti.Camera16(ti.BayerPattern.RGGB, moving_alpha=0.3, device=dev)
    isp.tonemap_reinhard([isp.load_packed12(f) for f in fr[0]], gamma=0.6)
    for step, call in enumerate((lambda f: isp.tonemap_reinhard([isp.load_packed12(x) for x in f], gamma=0.6),
                                 lambda f: isp.update_metering([isp.load_packed12(x) for x in f]),
                                 lambda f: isp.process_packed12(f, gamma=0.6)), start=1):
        old, snap = isp.metrics, isp.metrics.clone()
        call(fr[step])
        torch.cuda.synchronize()
        assert isp.metrics is not old and isp.metrics.data_ptr() != old.data_ptr(), step
        assert torch.equal(old, snap), f"call {step} wrote into the previous metrics tensor"
        assert not torch.equal(isp.metrics, snap), step


def test_isp_process_packed12_against_the_oracle(ti, dev):
    """The same call against the oracle's load -> update_metering -> reinhard_isp (camera_isp.py:333-340,376-385,177-218)."""
    from oracle import c_oracle
    H, W, n = 96, 1024, 3
    isp = ti.Camera16(ti.BayerPattern.RGGB, moving_alpha=0.3, device=dev)
    st = c_oracle.IspState(0.3)
    for group in range(2):
        packs = [natural_packed12(np.random.default_rng(1100 + 10 * group + k), H, W, dark=0.02 * k) for k in range(n)]
        outs, images = isp.process_packed12([torch.from_numpy(p).to(dev) for p in packs], gamma=0.6, keep_images=True)
        refs = [O.isp_load_packed12(p, "f16") for p in packs]
        m = st.update_metering(refs)
        assert_close(isp.metrics.cpu().numpy(), m, f"group {group}: metrics", rel=2e-5)
        for k in range(n):
            ref_u8, ref_after = c_oracle.reinhard_isp(refs[k], m, gamma=0.6)
            assert_close(outs[k].cpu().numpy(), ref_u8, f"group {group} camera {k}: u8")
            assert_close(images[k].cpu().numpy(), ref_after, f"group {group} camera {k}: p")


def test_isp_process_packed12_falls_back_to_the_two_calls(ti, dev):
    """What the camera-group kernel does not take - Camera32, a resize, an orientation transform, another metering stride -
    goes through load_packed12_batch + tonemap_reinhard: same results as the two calls, whatever the path."""
    H, W, n = 64, 512, 2
    frames = [torch.from_numpy(natural_packed12(np.random.default_rng(1200 + k), H, W, dark=0.03 * k)).to(dev) for k in range(n)]
    for cam, extra in (("Camera32", {}), ("Camera16", dict(resize_width=256)), ("Camera16", dict(transform=ti.ImageTransform.rotate_90)),
                       ("Camera16", dict(metering_stride=4)), ("Camera16", dict(ids=True))):
        ids = extra.pop("ids", False)                     # the IDS byte layout (packed.py:37-44): a call argument
        a = getattr(ti, cam)(ti.BayerPattern.RGGB, device=dev, **extra)
        b = getattr(ti, cam)(ti.BayerPattern.RGGB, device=dev, **extra)
        outs, images = a.process_packed12(frames, gamma=0.6, keep_images=True, ids_format=ids)
        want_images = [b.load_packed12(f, ids) for f in frames]
        want = b.tonemap_reinhard(want_images, gamma=0.6)
        assert torch.equal(a.metrics, b.metrics), (cam, extra)
        for k in range(n):
            bits = torch.int16 if cam == "Camera16" else torch.int32             # (bit views: p may hold NaN)
            assert torch.equal(outs[k], want[k]) and torch.equal(images[k].view(bits), want_images[k].view(bits)), (cam, extra, k)


def test_isp_process_packed12_timeout_is_reported(ti, dev):
    """The camera-group kernel's wait for max_out with a budget of one poll: the first block to look gives up - the camera's
    fault word, the device's mailbox word, TonemapTimeout at the next call - nothing hangs, and the call after is clean."""
    from taichi_image_amd import _native
    from taichi_image_amd.camera_isp import TonemapTimeout
    L = _native.lib()
    H, W = 768, 1024
    frames = [torch.from_numpy(natural_packed12(np.random.default_rng(1300 + k), H, W)).to(dev) for k in range(3)]
    isp = ti.Camera16(ti.BayerPattern.RGGB, device=dev)
    good = isp.process_packed12(frames, gamma=0.6)
    torch.cuda.synchronize()
    assert L.mi_isp_camera_group_faults(1) == 0
    L.mi_isp_camera_group_set_poll_limit(1)
    try:
        isp.process_packed12(frames, gamma=0.6)
        torch.cuda.synchronize()
    finally:
        L.mi_isp_camera_group_set_poll_limit(0)
    assert L.mi_isp_camera_group_faults(0) != 0
    ws = _native.workspace(H, W, dev, slots=len(frames) + 1)
    off, per = int(L.mi_isp_workspace_error_offset(H, W)), int(L.mi_isp_workspace_bytes(H, W))
    words = [int(ws[i * per + off:i * per + off + 4].view(torch.int32).item()) for i in range(len(frames))]
    assert any(words), "no camera's fault word was set"
    import ctypes
    n_failed = ctypes.c_int(0)
    assert L.mi_isp_workspace_check(ws.data_ptr(), len(frames), H, W, None, ctypes.byref(n_failed), _native.stream_ptr(dev)) == 0
    assert n_failed.value == sum(1 for w in words if w)
    with pytest.raises(TonemapTimeout):
        isp.process_packed12(frames, gamma=0.6)
    isp2 = ti.Camera16(ti.BayerPattern.RGGB, device=dev)
    again = isp2.process_packed12(frames, gamma=0.6)
    torch.cuda.synchronize()
    assert all(torch.equal(x, y) for x, y in zip(again, good)) and L.mi_isp_camera_group_faults(0) == 0


def test_resident_grids_of_both_kinds_share_one_order(ti, dev):
    """A one-launch metering grid and a whole-frame grid on two streams: each needs all its blocks resident, so the library
    puts them in ONE order (round 3 kept two, and the two kinds could hold half of the chip each until their budgets ran
    out).  Results equal those of the same work on one stream; no fault word, no mailbox."""
    from taichi_image_amd import _native
    from taichi_image_amd.pipeline import pipeline12_reinhard
    L = _native.lib()
    H, W = 768, 1024
    frames = [torch.from_numpy(natural_packed12(np.random.default_rng(600 + k), H, W, dark=0.03 * k)).to(dev) for k in range(6)]
    big = torch.from_numpy(natural_packed12(np.random.default_rng(7), 1536, 2048)).to(dev)
    cam = ti.Camera16(ti.BayerPattern.RGGB, moving_alpha=0.3, device=dev)
    imgs = [cam.load_packed12(f) for f in frames]
    cam.update_metering(imgs)
    want_m = cam.metrics.clone()
    want_o = pipeline12_reinhard(big, whole_frame=True).clone()
    torch.cuda.synchronize()
    assert L.mi_isp_whole_frame_faults(1) == 0 and L.mi_isp_metering_faults(1) == 0
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    cams = [ti.Camera16(ti.BayerPattern.RGGB, moving_alpha=0.3, device=dev) for _ in range(8)]
    outs = []
    for k in range(8):
        with torch.cuda.stream(s1):
            cams[k].update_metering(imgs)
        with torch.cuda.stream(s2):
            outs.append(pipeline12_reinhard(big, whole_frame=True))
    torch.cuda.synchronize()
    assert L.mi_isp_whole_frame_faults(1) == 0 and L.mi_isp_metering_faults(1) == 0
    for k in range(8):
        assert torch.equal(cams[k].metrics, want_m) and torch.equal(outs[k], want_o), k


@pytest.mark.parametrize("cam", ["Camera16", "Camera32"])
def test_isp_tonemap_reinhard_yuv420_fused(ti, rng, dev, cam):
    """The fused second pass + YUV 4:2:0 conversion equals converting the u8 outputs of tonemap_reinhard, bit
    for bit, over two calls (rolling metering), including the in-place side effect on the inputs; and the
    fallback (orientation transform set) gives the conversion of the transformed outputs."""
    from taichi_image_amd import color
    H, W = 96, 160
    frames = [torch.from_numpy(natural_packed12(rng, H, W)).to(dev) for _ in range(3)]
    a = getattr(ti, cam)(ti.BayerPattern.RGGB, moving_alpha=0.2, device=dev)
    b = getattr(ti, cam)(ti.BayerPattern.RGGB, moving_alpha=0.2, device=dev)
    for call in range(2):
        ia = [a.load_packed12(f) for f in frames]
        ib = [b.load_packed12(f) for f in frames]
        want = [color.rgb_yuv420_image(o) for o in a.tonemap_reinhard(ia, gamma=0.6, intensity=1.2)]
        got = b.tonemap_reinhard_yuv420(ib, gamma=0.6, intensity=1.2)
        for g, w_, x, y in zip(got, want, ia, ib):
            assert g.shape == (H * 3 // 2, W) and g.dtype == torch.uint8
            assert torch.equal(g, w_), f"call {call}"
            assert torch.equal(x, y), "inputs overwritten with the same p"
    c = getattr(ti, cam)(ti.BayerPattern.RGGB, device=dev, transform=ti.ImageTransform.flip_horiz)
    d = getattr(ti, cam)(ti.BayerPattern.RGGB, device=dev, transform=ti.ImageTransform.flip_horiz)
    ic, id_ = [c.load_packed12(frames[0])], [d.load_packed12(frames[0])]
    assert torch.equal(d.tonemap_reinhard_yuv420(id_, gamma=0.6)[0], color.rgb_yuv420_image(c.tonemap_reinhard(ic, gamma=0.6)[0]))


@pytest.mark.gpu
@pytest.mark.parametrize("cam,shape,rw", [("Camera16", (200, 512), 0), ("Camera32", (72, 264), 0), ("Camera16", (192, 512), 256),
                                          ("Camera16", (70, 200), 0)])
def test_load_packed_leaves_the_metering_subsample(ti, rng, dev, cam, shape, rw):
    """load_packed12 hands update_metering the stride-8 subsample it would otherwise gather (camera_isp.py:168-170): the
    dense copy equals image[::8, ::8] bit for bit (written by the streaming load kernel itself; frames it does not take,
    and resized loads, carry no subsample), metering on it gives the bits of metering on the images, and an image that was
    written to through torch afterwards falls back to the gather.  The C entry's gather fallback is checked directly."""
    H, W = shape
    packed = [natural_packed12(np.random.default_rng(40 + k), H, W) for k in range(3)]
    frames = [torch.from_numpy(p).to(dev) for p in packed]

    def fresh():
        return getattr(ti, cam)(ti.BayerPattern.RGGB, moving_alpha=0.3, resize_width=rw, device=dev)
    a, b = fresh(), fresh()
    for step in range(2):
        imgs = [a.load_packed12(f) for f in frames]
        fused = rw == 0 and W % 8 == 0
        for im in imgs:
            assert hasattr(im, "_mi_metering_sub") == fused        # only when the load kernel itself writes it
            if fused:
                sub, stride, ver = im._mi_metering_sub
                assert stride == 8 and torch.equal(sub, im[::8, ::8]), "subsample differs from image[::8, ::8]"
        plain = [im.clone() for im in imgs]                         # no tag: the strided gather
        a.update_metering(imgs)
        b.update_metering(plain)
        assert torch.equal(a.metrics, b.metrics), (a.metrics, b.metrics)
    # the C entry for a frame the fused path does not take (stride 4): the gather behind the load fills the buffer
    from taichi_image_amd import _native
    if rw == 0 and cam == "Camera16":
        rgb = torch.empty((H, W, 3), dtype=torch.float16, device=dev)
        sub4 = torch.zeros(((H + 3) // 4, (W + 3) // 4, 3), dtype=torch.float16, device=dev)
        _native.check(_native.lib().mi_isp_load_packed_metered(frames[0].data_ptr(), rgb.data_ptr(), H, W, 12, 0, 0, None, ti.types.f16.code,
                                                               H, W, 0.0, sub4.data_ptr(), 4, _native.stream_ptr(dev)))
        assert torch.equal(sub4, rgb[::4, ::4]) and torch.equal(rgb, a.load_packed12(frames[0]))
    imgs = [a.load_packed12(f) for f in frames]
    imgs[1].mul_(0.5)                                               # torch wrote to it: the tag is stale and must not be used
    plain = [im.clone() for im in imgs]
    a.update_metering(imgs)
    b.update_metering(plain)
    assert torch.equal(a.metrics, b.metrics)


@pytest.mark.parametrize("cam,shape,rw,n", [("Camera16", (200, 512), 0, 6), ("Camera32", (72, 264), 0, 3), ("Camera16", (192, 512), 256, 11),
                                            ("Camera16", (70, 200), 0, 2), ("Camera16", (96, 256), 77, 9)])
def test_load_packed_batch_equals_single_loads(ti, rng, dev, cam, shape, rw, n):
    """ISP.load_packed12_batch (extension; mi_isp_load_packed_batch: the cameras of a group in one launch per 8) gives the
    bits of a load_packed12 per camera - full size with the metering subsample, the fused resize, more than 8 frames (two
    launches), and frames the streaming kernels do not take (W % 8 != 0: loaded one by one)."""
    H, W = shape
    packed = [natural_packed12(np.random.default_rng(60 + k), H, W, pattern=O.GRBG) for k in range(n)]
    frames = [torch.from_numpy(p).to(dev) for p in packed]
    isp = getattr(ti, cam)(ti.BayerPattern.GRBG, resize_width=rw, correct_colors=True, device=dev)
    single = [isp.load_packed12(f) for f in frames]
    batch = isp.load_packed12_batch(frames)
    assert len(batch) == n
    for k in range(n):
        assert torch.equal(batch[k], single[k]), f"frame {k} differs"
        assert hasattr(batch[k], "_mi_metering_sub") == hasattr(single[k], "_mi_metering_sub")
        if hasattr(batch[k], "_mi_metering_sub"):
            assert torch.equal(batch[k]._mi_metering_sub[0], single[k]._mi_metering_sub[0])
    want = O.isp_load_packed12(packed[n - 1], "f16" if cam == "Camera16" else "f32", pattern=O.GRBG, resize_width=rw,
                               correct_colors=O.isp_color_matrix(True, O.DEFAULT_WB, O.DEFAULT_CC))
    assert_exact(batch[n - 1].cpu().numpy(), want, "last frame of the batch against the oracle")
    a = getattr(ti, cam)(ti.BayerPattern.GRBG, resize_width=rw, correct_colors=True, device=dev)
    b = getattr(ti, cam)(ti.BayerPattern.GRBG, resize_width=rw, correct_colors=True, device=dev)
    ua = a.tonemap_reinhard(a.load_packed12_batch(frames), gamma=0.7)
    ub = b.tonemap_reinhard([b.load_packed12(f) for f in frames], gamma=0.7)
    assert torch.equal(a.metrics, b.metrics) and all(torch.equal(x, y) for x, y in zip(ua, ub))
    assert isp.load_packed12_batch([]) == []


# ---- re-use of images the library has tone-mapped in place (camera_isp.py:211 then :376-403 again) -----------------
@pytest.mark.gpu
@pytest.mark.parametrize("cam", ["Camera16", "Camera32"])
@pytest.mark.parametrize("sequence", ["reinhard_twice", "reinhard_then_linear", "kernel_then_metering"])
def test_isp_reuse_of_tonemapped_images(ti, dev, cam, sequence):
    """(200, 512): the load kernel leaves the metering subsample (W % 8 == 0, no resize) - the case that went stale."""
    from oracle import c_oracle
    if not c_oracle.available():
        pytest.skip("oracle/liborc_isp.so not built")
    frames = [torch.from_numpy(natural_packed12(np.random.default_rng(70 + k), 200, 512, dark=0.05 * k)).to(dev) for k in range(3)]
    reuse_case(ti, dev, cam, frames, sequence)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["inference_mode", "no_grad"])
@pytest.mark.parametrize("cam", ["Camera16", "Camera32"])
def test_isp_under_inference_mode_and_no_grad(ti, dev, cam, mode):
    """The reference's bench runs inside torch.inference_mode() (bench/camera_isp.py:53): tensors created there have no
    version counter.  load_packed12 (single and batched) -> tonemap_reinhard twice must work and agree with the oracle."""
    from oracle import c_oracle
    if not c_oracle.available():
        pytest.skip("oracle/liborc_isp.so not built")
    packed = [natural_packed12(np.random.default_rng(90 + k), 200, 512, dark=0.04 * k) for k in range(2)]
    with getattr(torch, mode)():
        frames = [torch.from_numpy(p).to(dev) for p in packed]
        isp = getattr(ti, cam)(ti.BayerPattern.RGGB, moving_alpha=0.5, device=dev)
        st = c_oracle.IspState(0.5)
        imgs = isp.load_packed12_batch(frames[:1]) + [isp.load_packed12(frames[1])]
        if mode == "inference_mode":
            assert all(torch.is_inference(im) and not hasattr(im, "_mi_metering_sub") for im in imgs)
        for step in range(2):
            cur = [im.cpu().numpy() for im in imgs]
            outs = isp.tonemap_reinhard(imgs, gamma=0.6)
            m = st.update_metering(cur)
            assert_close(isp.metrics.cpu().numpy(), m, f"metrics {step}", rel=2e-5)
            for k in range(2):
                ref_u8, ref_after = c_oracle.reinhard_isp(cur[k], m, gamma=0.6)
                assert_close(outs[k].cpu().numpy(), ref_u8, f"u8 step {step} img {k}")
                assert_close(imgs[k].cpu().numpy(), ref_after, f"p step {step} img {k}")


@pytest.mark.gpu
def test_in_place_write_is_visible_to_torch(ti, dev):
    """What the library writes through raw pointers moves the tensor's version counter (views included)."""
    frame = torch.from_numpy(natural_packed12(np.random.default_rng(5), 64, 256)).to(dev)
    isp = ti.Camera16(ti.BayerPattern.RGGB, moving_alpha=1.0, device=dev)
    im = isp.load_packed12(frame)
    view = im[:8]
    v0 = im._version
    isp.tonemap_reinhard([im])
    assert im._version > v0 and view._version == im._version


# ---- the HIP path against the COMMITTED fixtures (tests/golden/golden_small.npz) -----------------------------------
@pytest.mark.gpu
def test_hip_path_against_the_committed_fixtures(ti, dev):
    """Every vector of golden_small.npz (tests/golden/make_golden.py) recomputed by the HIP kernels and compared with the
    file itself - not with the live oracle: an edit that moved oracle and kernel together would still fail here."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_small.npz"))
    enc = g["dec_in"]
    assert_exact(ti.packed.decode12(enc, dtype=ti.types.u16), g["dec_std_u16"], "dec_std_u16")
    assert_exact(ti.packed.decode12(enc, dtype=ti.types.u16, ids_format=True), g["dec_ids_u16"], "dec_ids_u16")
    assert_exact(ti.packed.decode12(enc, dtype=ti.types.f16, scaled=True), g["dec_std_f16s"], "dec_std_f16s")
    assert_exact(ti.packed.decode12(enc, dtype=ti.types.f32, scaled=True, ids_format=True), g["dec_ids_f32s"], "dec_ids_f32s")
    assert_exact(ti.packed.decode16(enc, dtype=ti.types.f16, scaled=True), g["dec16_f16s"], "dec16_f16s")
    for p in range(4):
        assert_exact(ti.bayer.bayer_to_rgb(g["cfa_u16"], pat(ti, p)), g[f"rgb_u16_p{p}"], f"rgb_u16_p{p}")
        assert_exact(ti.bayer.bayer_to_rgb(g["cfa_f16"], pat(ti, p)), g[f"rgb_f16_p{p}"], f"rgb_f16_p{p}")
    ccm = O.isp_color_matrix(True, O.DEFAULT_WB, O.DEFAULT_CC)          # (plain numpy arithmetic of camera_isp.py:360-369)
    assert_exact(ti.bayer.bayer_to_rgb(g["cfa_f16"], pat(ti, 0), correct_colors=ccm), g["rgb_f16_ccm"], "rgb_f16_ccm")
    from taichi_image_amd.pipeline import pipeline12_reinhard
    for p, ids in ((0, False), (2, True)):
        tag = f"p{p}{'i' if ids else 's'}"
        raw = torch.from_numpy(g[f"packed_{tag}"]).to(dev)
        for wf in ((False, True) if not ids else (False,)):              # the multi-pass chain and the whole-frame kernel
            got = pipeline12_reinhard(raw, pat(ti, p), ids, whole_frame=wf).cpu().numpy()
            assert_close(got, g[f"pipe_f16_{tag}"], f"pipe_f16_{tag} whole_frame={wf}")
            got8 = pipeline12_reinhard(raw, pat(ti, p), ids, dtype=ti.types.u8, gamma=0.6, whole_frame=wf).cpu().numpy()
            assert_close(got8, g[f"pipe_u8_{tag}"], f"pipe_u8_{tag} whole_frame={wf}")
    isp = ti.Camera16(ti.BayerPattern.RGGB, moving_alpha=0.1, resize_width=40, device=dev)
    im = isp.load_packed12(torch.from_numpy(g["packed_p0s"]).to(dev))
    assert_exact(im.cpu().numpy(), g["isp_img_f16"], "isp_img_f16")
    isp.update_metering([im, im])
    assert_close(isp.metrics.cpu().numpy(), g["isp_metrics_1"], "isp_metrics_1", rel=2e-5)
    lin = isp.tonemap_linear([im], gamma=0.8)[0]
    assert_close(isp.metrics.cpu().numpy(), g["isp_metrics_2"], "isp_metrics_2", rel=2e-5)
    assert_close(lin.cpu().numpy(), g["isp_linear_u8"], "isp_linear_u8")
    out = isp.tonemap_only(im, isp.metrics, 0.6, 1.0, 1.0, 0.0)
    assert_close(out.cpu().numpy(), g["isp_u8"], "isp_u8")
    assert_close(im.cpu().numpy(), g["isp_after"], "isp_after")
    assert_exact(ti.interpolate.resize_bilinear(g["scene_f32"], (30, 20), 0.46875), g["resize_f32"], "resize_f32")
