"""The C/OpenMP restatement (oracle/isp_oracle.c) against the NumPy restatement: two independent
implementations of the same specification must agree -- bit-exactly for unpack, f16 rounding and
demosaic, within the fp tolerance for the libm-dependent tonemap."""
import numpy as np
import pytest

from oracle import c_oracle, isp_oracle as O
from tests.util import assert_close, assert_exact, natural_packed12

pytestmark = pytest.mark.skipif(not c_oracle.available(), reason="oracle/liborc_isp.so not built (run build())")


def test_tables_and_f16_rounding(rng):
    assert np.array_equal(c_oracle.bayer_kernels(), O.BAYER_KERNELS)
    x = np.concatenate([rng.random(20000, dtype=np.float32) * 2 - 1, (rng.random(2000) * 1e-4).astype(np.float32),
                        (rng.random(2000) * 1e-7).astype(np.float32),
                        np.array([0, 1, 65504, 65519.9, 65520, 1e6, 2.0 ** -24, 2.0 ** -25, 6.1e-5, np.inf, -np.inf], np.float32)])
    with np.errstate(over="ignore"):
        want = x.astype(np.float16).view(np.uint16)
    got = np.array([c_oracle.lib().orc_f32_to_f16_bits(float(v)) for v in x], np.uint16)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("ids", [False, True])
def test_decode12(rng, ids):
    enc = rng.integers(0, 256, 3 * 4001).astype(np.uint8)
    for work in ("f16", "f32"):
        assert_exact(c_oracle.decode12_scaled(enc, ids, work), O.decode12(enc, work, True, ids).astype(np.float32))


@pytest.mark.parametrize("p", [0, 1, 2, 3])
def test_demosaic_bit_exact(rng, p):
    cfa = rng.random((38, 54), dtype=np.float32)
    assert_exact(c_oracle.demosaic(cfa, p), O.bayer_to_rgb(cfa, p))
    h = cfa.astype(np.float16)
    assert_exact(c_oracle.demosaic(h.astype(np.float32), p, round_f16=True).astype(np.float16), O.bayer_to_rgb(h, p))
    ccm = O.isp_color_matrix(True, O.DEFAULT_WB, O.DEFAULT_CC)
    assert_exact(c_oracle.demosaic(cfa, p, ccm=ccm), O.bayer_to_rgb(cfa, p, ccm))
    u16 = rng.integers(0, 65536, (12, 20)).astype(np.uint16)
    got = c_oracle.demosaic(u16.astype(np.float32), p, in_scale=65535.0)
    assert_exact(O.cast_out(got * np.float32(65535), "u16"), O.bayer_to_rgb(u16, p))


@pytest.mark.parametrize("out", ["f16", "u8", "f32"])
@pytest.mark.parametrize("kw", [dict(), dict(gamma=0.6, intensity=1.5, light_adapt=0.7, color_adapt=0.4)])
def test_pipeline(rng, out, kw):
    packed = natural_packed12(rng, 64, 96)
    work = "f32" if out == "f32" else "f16"
    got = c_oracle.pipeline12_reinhard(packed, work=work, out=out, **kw)
    assert_close(got, O.pipeline12_reinhard(packed, work=work, out=out, **kw), f"C vs NumPy pipeline {out}")


@pytest.mark.parametrize("work", ["f16", "f32"])
def test_isp_stateful_path(rng, work):
    """camera_isp.py:142-227 in C against the NumPy restatement: the 9-vector over three steps (moving
    average), the u8 outputs and the in-place write-back of Reinhard, the linear map."""
    st_c, st_n = c_oracle.IspState(0.3), O.IspState(0.3)
    for step in range(3):
        imgs = [O.isp_load_packed12(natural_packed12(rng, 80, 96, dark=0.05 * step), work) for _ in range(3)]
        mc, mn = st_c.update_metering(imgs), st_n.update_metering(imgs)
        assert_close(mc, mn, f"metering step {step}", rel=2e-6)
        for kw in (dict(gamma=0.6), dict(gamma=1.0, intensity=0.7, light_adapt=0.8, color_adapt=0.3)):
            for im in imgs:
                u8_c, after_c = c_oracle.reinhard_isp(im, mn, **kw)
                u8_n, after_n = O.reinhard_isp(im, mn, **kw)
                assert_close(u8_c, u8_n, "reinhard u8")
                assert_close(after_c, after_n, "reinhard write-back")
        assert_close(c_oracle.linear_isp(imgs[0], mn, 0.8), O.linear_isp(imgs[0], mn, 0.8), "linear")
