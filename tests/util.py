"""Shared helpers for the parity tests."""
import numpy as np

from oracle import isp_oracle as O

NP = O.NP_DTYPE


def bits(a: np.ndarray) -> np.ndarray:
    """Bit view for exact comparison of float arrays (NaN-safe, -0 != +0)."""
    a = np.ascontiguousarray(a)
    if a.dtype == np.float16:
        return a.view(np.uint16)
    if a.dtype == np.float32:
        return a.view(np.uint32)
    return a


def assert_exact(got, ref, what=""):
    got, ref = np.asarray(got), np.asarray(ref)
    assert got.shape == ref.shape, f"{what}: shape {got.shape} != {ref.shape}"
    assert got.dtype == ref.dtype, f"{what}: dtype {got.dtype} != {ref.dtype}"
    bad = bits(got) != bits(ref)
    if bad.any():
        idx = tuple(int(i[0]) for i in np.nonzero(bad))
        raise AssertionError(f"{what}: {bad.sum()} / {bad.size} elements differ, first at {idx}: "
                             f"got {got[idx]!r} want {ref[idx]!r}")


def assert_close(got, ref, what="", rel=1e-4, max_bad_frac=0.0):
    """The parity contract for floating-point stages: |got - ref| <= rel*|ref| + one unit of the
    output type (1 LSB for integer outputs, 1 ulp for f16/f32 outputs).  NaNs must coincide."""
    got, ref = np.asarray(got), np.asarray(ref)
    assert got.shape == ref.shape, f"{what}: shape {got.shape} != {ref.shape}"
    assert got.dtype == ref.dtype, f"{what}: dtype {got.dtype} != {ref.dtype}"
    g, r = got.astype(np.float64), ref.astype(np.float64)
    nan_g, nan_r = np.isnan(g), np.isnan(r)
    assert np.array_equal(nan_g, nan_r), f"{what}: NaN pattern differs ({nan_g.sum()} vs {nan_r.sum()})"
    if ref.dtype == np.float16:
        unit = np.maximum(np.abs(r), 2.0 ** -14) * 2.0 ** -10   # one f16 ulp (>= spacing at |r|)
    elif ref.dtype == np.float32:
        unit = np.maximum(np.abs(r), 1e-30) * 2.0 ** -22 + 1e-7
    else:
        unit = 1.0
    err = np.abs(g - r)
    tol = rel * np.abs(r) + unit
    bad = (err > tol) & ~nan_r
    frac = bad.mean() if bad.size else 0.0
    if frac > max_bad_frac:
        idx = tuple(int(i[0]) for i in np.nonzero(bad))
        raise AssertionError(f"{what}: {bad.sum()} / {bad.size} outside tolerance (max err {err[~nan_r].max():.4g}), "
                             f"first at {idx}: got {got[idx]!r} want {ref[idx]!r}")
    return float(err[~nan_r].max()) if (~nan_r).any() else 0.0


def random_cfa(rng, H, W, dtype):
    if dtype == "u8":
        return rng.integers(0, 256, (H, W)).astype(np.uint8)
    if dtype == "u16":
        return rng.integers(0, 65536, (H, W)).astype(np.uint16)
    x = rng.random((H, W), dtype=np.float32)
    return x.astype(NP[dtype])


def natural_packed12(rng, H, W, pattern=O.RGGB, ids_format=False, dark=0.0):
    """A smooth-plus-noise scene, mosaiced and packed (12 bit)."""
    r = np.arange(H)[:, None] / max(H, 1)
    c = np.arange(W)[None, :] / max(W, 1)
    base = 0.1 + 0.8 * (0.5 + 0.5 * np.sin(6.0 * r + 1.0)) * (0.5 + 0.5 * np.cos(9.0 * c))
    img = np.stack([np.clip(base * g + rng.normal(0, 0.03, (H, W)) - dark, 0, 1) for g in (1.0, 0.8, 0.6)], -1)
    cfa = O.rgb_to_bayer(img.astype(np.float32), pattern)
    v12 = np.rint(cfa.astype(np.float64) * 4095).astype(np.uint16)
    return O.encode12(v12, ids_format=ids_format)


def reuse_case(ti, dev, cam, frames, sequence, alpha=0.3):
    """One list of images through two calls.  The first Reinhard call overwrites every image with its mapped values p
    (camera_isp.py:211); the second call must meter the MUTATED images (camera_isp.py:168-175 reads the tensors it is
    given), not the subsample the load kernel left behind.  The oracle of the second call consumes the device's images
    as they stand after the first (their own parity is asserted first)."""
    import torch
    from oracle import c_oracle
    isp = getattr(ti, cam)(ti.BayerPattern.RGGB, moving_alpha=alpha, device=dev)
    st = c_oracle.IspState(alpha)
    imgs = [isp.load_packed12(f) for f in frames]
    refs = [im.cpu().numpy() for im in imgs]
    if sequence == "kernel_then_metering":
        # the static kernel with metrics of the caller's own, as ISP.tonemap_only does (camera_isp.py:387-390)
        isp.update_metering(imgs)
        m1 = st.update_metering(refs)
        assert_close(isp.metrics.cpu().numpy(), m1, "metrics 1", rel=2e-5)
        for k, im in enumerate(imgs):
            out = torch.empty(im.shape, dtype=torch.uint8, device=dev)
            type(isp).reinhard_kernel(im, out, isp.metrics, 0.6, 1.0, 1.0, 0.0)
            ref_u8, ref_after = c_oracle.reinhard_isp(refs[k], m1, gamma=0.6)
            assert_close(out.cpu().numpy(), ref_u8, f"u8 img {k}")
            assert_close(im.cpu().numpy(), ref_after, f"in-place p img {k}")
            assert not hasattr(im, "_mi_metering_sub"), "a subsample taken before the write survived it"
        mut = [im.cpu().numpy() for im in imgs]
        isp.update_metering(imgs)
        assert_close(isp.metrics.cpu().numpy(), st.update_metering(mut), "metrics after ISP.reinhard_kernel", rel=2e-5)
        return
    outs = isp.tonemap_reinhard(imgs, gamma=0.6)
    m1 = st.update_metering(refs)
    assert_close(isp.metrics.cpu().numpy(), m1, "metrics 1", rel=2e-5)
    for k in range(len(imgs)):
        ref_u8, ref_after = c_oracle.reinhard_isp(refs[k], m1, gamma=0.6)
        assert_close(outs[k].cpu().numpy(), ref_u8, f"u8 1 img {k}")
        assert_close(imgs[k].cpu().numpy(), ref_after, f"in-place p 1 img {k}")
    mut = [im.cpu().numpy() for im in imgs]
    assert any(not np.array_equal(a, b) for a, b in zip(mut, refs)), "the first call did not write p back"
    if sequence == "reinhard_twice":
        outs2 = isp.tonemap_reinhard(imgs, gamma=0.8)
        m2 = st.update_metering(mut)
        assert_close(isp.metrics.cpu().numpy(), m2, "metrics 2", rel=2e-5)
        for k in range(len(imgs)):
            ref_u8, ref_after = c_oracle.reinhard_isp(mut[k], m2, gamma=0.8)
            assert_close(outs2[k].cpu().numpy(), ref_u8, f"u8 2 img {k}")
            assert_close(imgs[k].cpu().numpy(), ref_after, f"in-place p 2 img {k}")
    else:
        assert sequence == "reinhard_then_linear"
        outs2 = isp.tonemap_linear(imgs, gamma=0.8)
        m2 = st.update_metering(mut)
        assert_close(isp.metrics.cpu().numpy(), m2, "metrics 2", rel=2e-5)
        for k in range(len(imgs)):
            assert_close(outs2[k].cpu().numpy(), c_oracle.linear_isp(mut[k], m2, 0.8), f"linear 2 img {k}")
    # the stale subsample would have given the metrics of the un-mutated images: make sure the two differ at all
    stale = c_oracle.IspState(alpha)
    stale.update_metering(refs)
    assert not np.allclose(stale.update_metering(refs), m2, rtol=1e-3), "the sequence does not tell stale from fresh"
