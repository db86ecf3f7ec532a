"""GPU parity at the BASELINE sizes for the stateful ISP leg (camera_isp.py:142-227,376-413) and the
general (non-unit bounds) path of the fused config-2 chain.

The small-size tests of test_gpu_parity.py run these kernels with one block per image; here
`metering_kernel` (up to 256 blocks per image), the batched `rgb_pass_kernel<PM_ISP_RH_P1/P2>`
(grid.y = image, per-image max_out folding) and the 64-image chunking meet the C oracle
(oracle/isp_oracle.c, pinned against the NumPy restatement by tests/test_c_oracle.py) with many
blocks, six images of differing brightness and three consecutive calls (moving average).

The oracle consumes the image the device loader produced (its own parity is bit-exact and tested
in test_gpu_parity.py), copied to the host BEFORE the tonemap mutates it in place.
"""
import numpy as np
import pytest
import torch

from oracle import c_oracle, isp_oracle as O
from tests.util import assert_close, natural_packed12
from taichi_image_amd.synthetic import mosaic_rggb, pack12, synthetic_scene

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not c_oracle.available(), reason="oracle/liborc_isp.so not built")]


@pytest.fixture(scope="module")
def ti():
    import taichi_image_amd as t
    return t


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def scenes():
    """Six 4096x3072 scenes (SURVEY 8(d) generator), kept as f32 CFA in [0, 1]."""
    return [mosaic_rggb(synthetic_scene(k)) for k in range(6)]


def packed_from(cfa, gain=1.0, offset=0.0):
    """12-bit frame of gain * scene + offset (clipped): frames of differing brightness and bounds."""
    v12 = np.rint(np.clip(cfa.astype(np.float64) * gain + offset, 0, 1) * 4095).astype(np.uint16)
    return pack12(v12)


# brightness of the six cameras of one call, and the drift from call to call (moving average)
GAINS = [1.0, 0.55, 0.8, 0.3, 0.95, 0.7]
OFFSETS = [0.0, 0.02, 0.1, 0.0, 0.04, 0.15]


def load_group(isp, scenes, step, dev):
    imgs = []
    for k, cfa in enumerate(scenes):
        p = packed_from(cfa, GAINS[(k + step) % 6] * (1.0 - 0.12 * step), OFFSETS[(k + 2 * step) % 6])
        imgs.append(isp.load_packed12(torch.from_numpy(p).to(dev)))
    return imgs


@pytest.mark.parametrize("cam,resize_width", [("Camera16", 1920), ("Camera16", 0), ("Camera32", 0)])
def test_isp_reinhard_full_size_sequence(ti, dev, scenes, cam, resize_width):
    """Camera16(resize_width=1920) -> 1440x1920 and full-resolution Camera16 / Camera32: six images, three
    consecutive tonemap_reinhard(gamma=0.6) calls; metrics, u8 outputs and the in-place p against the C oracle."""
    isp = getattr(ti, cam)(ti.BayerPattern.RGGB, moving_alpha=0.3, resize_width=resize_width, device=dev)
    st = c_oracle.IspState(0.3)
    for step in range(3):
        imgs = load_group(isp, scenes, step, dev)
        refs = [im.cpu().numpy() for im in imgs]                 # before the in-place write-back
        assert refs[0].shape == ((1440, 1920, 3) if resize_width else (3072, 4096, 3))
        outs = isp.tonemap_reinhard(imgs, gamma=0.6)
        m = st.update_metering(refs)
        assert_close(isp.metrics.cpu().numpy(), m, f"{cam} metrics step {step}", rel=2e-5)
        for k, (o, im, r) in enumerate(zip(outs, imgs, refs)):
            ref_u8, ref_after = c_oracle.reinhard_isp(r, m, gamma=0.6)
            assert_close(o.cpu().numpy(), ref_u8, f"{cam} u8 step {step} img {k}")
            assert_close(im.cpu().numpy(), ref_after, f"{cam} in-place p step {step} img {k}")
        # the six images must not share a max_out: their outputs differ
        assert len({int(o.float().mean().item() * 1000) for o in outs}) > 1


@pytest.mark.parametrize("cam,resize_width", [("Camera16", 1920), ("Camera32", 0)])
def test_isp_linear_full_size(ti, dev, scenes, cam, resize_width):
    isp = getattr(ti, cam)(ti.BayerPattern.RGGB, moving_alpha=0.5, resize_width=resize_width, device=dev)
    st = c_oracle.IspState(0.5)
    for step in range(2):
        imgs = load_group(isp, scenes[:4], step, dev)
        refs = [im.cpu().numpy() for im in imgs]
        outs = isp.tonemap_linear(imgs, gamma=0.8)
        m = st.update_metering(refs)
        assert_close(isp.metrics.cpu().numpy(), m, f"{cam} metrics step {step}", rel=2e-5)
        for k, (o, r) in enumerate(zip(outs, refs)):
            assert_close(o.cpu().numpy(), c_oracle.linear_isp(r, m, 0.8), f"{cam} linear step {step} img {k}")


def test_isp_reinhard_yuv420_full_size(ti, dev, scenes):
    """The fused second pass + YUV conversion at 1440x1920, four images of differing brightness, two steps."""
    isp = ti.Camera16(ti.BayerPattern.RGGB, moving_alpha=0.4, resize_width=1920, device=dev)
    st = c_oracle.IspState(0.4)
    for step in range(2):
        imgs = load_group(isp, scenes[:4], step, dev)
        refs = [im.cpu().numpy() for im in imgs]
        outs = isp.tonemap_reinhard_yuv420(imgs, gamma=0.6)
        m = st.update_metering(refs)
        for k, (o, r) in enumerate(zip(outs, refs)):
            ref_u8, _ = c_oracle.reinhard_isp(r, m, gamma=0.6)
            got = o.cpu().numpy()
            want = O.rgb_yuv420(ref_u8)
            assert got.shape == want.shape
            # a u8 RGB value one LSB off (the tonemap tolerance) moves Y by at most one LSB and U/V by less
            d = np.abs(got.astype(np.int32) - want.astype(np.int32))
            assert d.max() <= 1, f"yuv420 step {step} img {k}: max |diff| {d.max()}"
            assert (d > 0).mean() < 0.02, f"yuv420 step {step} img {k}: {(d > 0).mean():.4f} of the bytes differ"


def test_isp_reinhard_70_images_cross_the_chunk(ti, dev, rng):
    """A 70-image list crosses the 64-image chunking of mi_isp_reinhard_batch / mi_isp_linear_batch."""
    isp = ti.Camera16(ti.BayerPattern.RGGB, moving_alpha=1.0, device=dev)
    packs = [natural_packed12(rng, 48, 64, dark=0.004 * k) for k in range(70)]
    imgs = [isp.load_packed12(torch.from_numpy(p).to(dev)) for p in packs]
    refs = [im.cpu().numpy() for im in imgs]
    lin = isp.tonemap_linear([im.clone() for im in imgs], gamma=0.9)
    m = c_oracle.IspState(1.0).update_metering(refs)
    for k in (0, 1, 63, 64, 65, 69):
        assert_close(lin[k].cpu().numpy(), c_oracle.linear_isp(refs[k], m, 0.9), f"linear img {k}")
    isp.metrics = None
    outs = isp.tonemap_reinhard(imgs, gamma=0.7)
    assert_close(isp.metrics.cpu().numpy(), m, "metrics of 70 images", rel=2e-5)
    for k in range(70):
        ref_u8, ref_after = c_oracle.reinhard_isp(refs[k], m, gamma=0.7)
        assert_close(outs[k].cpu().numpy(), ref_u8, f"u8 img {k}")
        assert_close(imgs[k].cpu().numpy(), ref_after, f"in-place p img {k}")


@pytest.mark.parametrize("out", ["f16", "u8"])
def test_pipeline12_4k_non_unit_bounds(ti, dev, scenes, out):
    """Config 2 at full size on a frame whose demosaiced bounds are NOT (0, 1) (scene scaled into [0.1, 0.8]):
    the general normalisation path of every pass, which no synthetic bench frame reaches, against the C oracle."""
    from taichi_image_amd.pipeline import pipeline12_reinhard
    packed = packed_from(scenes[2], 0.7, 0.1)
    ref = c_oracle.pipeline12_reinhard(packed, work="f16", out=out)
    cfa = c_oracle.decode12_scaled(packed, work="f16").reshape(3072, 4096)
    rgb = c_oracle.demosaic(cfa, 0, round_f16=True)
    assert rgb.min() > 0.0 and rgb.max() < 1.0, "the frame must not touch the clamp"
    got = pipeline12_reinhard(torch.from_numpy(packed).to(dev), dtype=getattr(ti.types, out), whole_frame=False).cpu().numpy()
    assert_close(got, ref, f"pipeline12 4K non-unit bounds -> {out}")


# ---- the whole-frame kernel (csrc/isp_mega.h): one persistent launch with grid barriers ------------------------
def _error_word(ti, H, W, dev):
    from taichi_image_amd import _native
    ws = _native.workspace(H, W, dev)
    off = int(_native.lib().mi_isp_workspace_error_offset(H, W))
    return int(ws[off:off + 4].view(torch.int32).item())


@pytest.mark.parametrize("shape", [(4, 8), (12, 512), (26, 520), (64, 512), (130, 1544), (3072, 4096)])
@pytest.mark.parametrize("out", ["f16", "u8"])
def test_whole_frame_kernel_matches_c_oracle(ti, dev, rng, shape, out):
    """Every frame shape class: one wave, partial bands, partial last row band, partial block, the full chip."""
    from taichi_image_amd.pipeline import pipeline12_reinhard, whole_frame_fits
    H, W = shape
    assert whole_frame_fits(H, W, getattr(ti.types, out))
    packed = natural_packed12(rng, H, W) if H < 3072 else packed_from(mosaic_rggb(synthetic_scene(7)))
    ref = c_oracle.pipeline12_reinhard(packed, work="f16", out=out)
    got = pipeline12_reinhard(torch.from_numpy(packed).to(dev), dtype=getattr(ti.types, out), whole_frame=True)
    torch.cuda.synchronize()
    assert _error_word(ti, H, W, dev) == 0, "a grid barrier of the whole-frame kernel timed out"
    assert_close(got.cpu().numpy(), ref, f"whole-frame {H}x{W} -> {out}")


@pytest.mark.parametrize("kw", [dict(gamma=0.6, intensity=1.5, light_adapt=0.7, color_adapt=0.4),
                                dict(gamma=2.2, intensity=0.5, light_adapt=1.0, color_adapt=0.0)])
def test_whole_frame_kernel_parameters_patterns_and_bounds(ti, dev, scenes, kw):
    """Non-default Reinhard parameters, every CFA pattern, a colour matrix, and a frame whose bounds are not (0, 1)
    (the in-kernel statistics fallback, phase B) at full size; repeated launches on one workspace."""
    from taichi_image_amd.pipeline import pipeline12_reinhard
    for p in range(4):
        packed = natural_packed12(np.random.default_rng(5 + p), 96, 1024, pattern=p)
        ccm = O.isp_color_matrix(True, O.DEFAULT_WB, O.DEFAULT_CC) if p == 2 else None
        ref = O.pipeline12_reinhard(packed, pattern=p, correct_colors=ccm, out="f16", **kw)
        got = pipeline12_reinhard(torch.from_numpy(packed).to(dev), pattern=ti.BayerPattern(p), correct_colors=ccm,
                                  whole_frame=True, **kw)
        assert_close(got.cpu().numpy(), ref, f"whole-frame pattern {p} {kw}")
    packed = packed_from(scenes[2], 0.7, 0.1)                      # bounds inside (0, 1)
    ref = c_oracle.pipeline12_reinhard(packed, work="f16", out="f16", **kw)
    frame = torch.from_numpy(packed).to(dev)
    for rep in range(3):                                           # the kernel re-arms its own barrier counters
        got = pipeline12_reinhard(frame, whole_frame=True, **kw)
        torch.cuda.synchronize()
        assert _error_word(ti, 3072, 4096, dev) == 0
        assert_close(got.cpu().numpy(), ref, f"whole-frame 4K non-unit bounds {kw} rep {rep}")


def test_whole_frame_kernel_streams_are_serialised(ti, dev, scenes):
    """Launches from two streams: the library chains them (two whole-frame grids must never share the chip)."""
    from taichi_image_amd.pipeline import pipeline12_reinhard
    packed = [packed_from(scenes[k]) for k in range(2)]
    frames = [torch.from_numpy(p).to(dev) for p in packed]
    refs = [c_oracle.pipeline12_reinhard(p, work="f16", out="f16") for p in packed]
    streams = [torch.cuda.Stream(dev) for _ in range(2)]
    outs = [None, None]
    torch.cuda.synchronize()
    for rep in range(4):
        for k in range(2):
            with torch.cuda.stream(streams[k]):
                outs[k] = pipeline12_reinhard(frames[k], whole_frame=True)
    torch.cuda.synchronize()
    from taichi_image_amd import _native
    for k in range(2):
        with torch.cuda.stream(streams[k]):
            assert _error_word(ti, 3072, 4096, dev) == 0
        assert_close(outs[k].cpu().numpy(), refs[k], f"stream {k}")


def test_whole_frame_kernel_shares_its_workspace(ti, dev, rng, scenes):
    """One workspace (same stream, same byte size) serves frames of different geometry, both chains and both kinds of
    bounds in any order: the barrier records carry the workspace's launch count, so records a smaller grid or the other
    chain left behind never count, and nothing has to be reset between launches."""
    from taichi_image_amd.pipeline import pipeline12_reinhard
    from taichi_image_amd import _native
    cases = {"4k": packed_from(scenes[0]), "4k-inside": packed_from(scenes[1], 0.7, 0.1),
             "small": natural_packed12(rng, 64, 512), "odd": natural_packed12(rng, 130, 1544)}
    refs = {k: c_oracle.pipeline12_reinhard(v, work="f16", out="f16") for k, v in cases.items()}
    frames = {k: torch.from_numpy(v).to(dev) for k, v in cases.items()}
    ws = {k: _native.workspace(v.shape[0], v.shape[1] * 2 // 3, dev).data_ptr() for k, v in cases.items()}
    assert len(set(ws.values())) == 1, "the cases are meant to share one workspace"
    order = [("4k", True), ("small", True), ("4k-inside", True), ("small", False), ("odd", True), ("4k", False),
             ("4k", True), ("4k-inside", True), ("odd", True), ("4k", True)]
    for step, (name, whole) in enumerate(order):
        got = pipeline12_reinhard(frames[name], whole_frame=whole)
        torch.cuda.synchronize()
        H, Wp = cases[name].shape
        assert _error_word(ti, H, Wp * 2 // 3, dev) == 0, f"step {step}: a grid barrier timed out"
        assert_close(got.cpu().numpy(), refs[name], f"step {step}: {name}, whole_frame={whole}")


def test_whole_frame_kernel_is_deterministic(ti, dev, scenes):
    """The barrier folds have a fixed order: every launch of a frame gives the same bits (scripts/wf_soak.py runs this for
    two million frames); a stale or torn barrier record would show up here as a differing output or an error word."""
    from taichi_image_amd.pipeline import pipeline12_reinhard
    frames = [torch.from_numpy(packed_from(scenes[0])).to(dev), torch.from_numpy(packed_from(scenes[1], 0.7, 0.1)).to(dev)]
    first = [pipeline12_reinhard(f, whole_frame=True).clone() for f in frames]
    outs = [torch.empty_like(o) for o in first]
    for it in range(400):
        k = it % 3 == 0
        pipeline12_reinhard(frames[k], out=outs[k], whole_frame=True)
        if it % 100 == 99:
            torch.cuda.synchronize()
            assert _error_word(ti, 3072, 4096, dev) == 0
            assert torch.equal(outs[0], first[0]) and torch.equal(outs[1], first[1]), f"launch {it}: output changed"


def test_whole_frame_kernel_refuses_what_it_cannot_hold(ti, dev):
    from taichi_image_amd.pipeline import pipeline12_reinhard, whole_frame_fits
    assert not whole_frame_fits(3072 + 12, 4096)            # one row band more than the chip holds
    assert not whole_frame_fits(64, 512, ti.types.f32)      # 4-byte outputs do not fit the staging slot
    big = torch.zeros((3084, 6144), dtype=torch.uint8, device=dev)
    with pytest.raises(RuntimeError, match="whole-frame"):
        pipeline12_reinhard(big, whole_frame=True)


def test_whole_frame_batch_launch_equals_single_launches(ti, dev, rng, scenes):
    """One launch walks through a batch (mi_isp_pipeline12_reinhard_whole_frame_batch): unit and non-unit frames mixed,
    more frames than one launch takes (64), every output bit-identical to a launch of its own and within tolerance of the
    C oracle; a small geometry with partial row bands too."""
    from taichi_image_amd.pipeline import pipeline12_reinhard, BatchPipeline
    packed = [packed_from(scenes[0]), packed_from(scenes[1], 0.7, 0.1), packed_from(scenes[2])]
    frames = [torch.from_numpy(p).to(dev) for p in packed]
    singles = [pipeline12_reinhard(f, whole_frame=True).clone() for f in frames]
    bp = BatchPipeline(5, 3072, 4096, dev, whole_frame=True)
    order = [0, 1, 2, 1, 0]
    outs = bp([frames[k] for k in order])
    assert bp.check() == []
    for i, k in enumerate(order):
        assert torch.equal(outs[i], singles[k]), f"batch frame {i} differs from its single launch"
    for k in range(2):
        assert_close(singles[k].cpu().numpy(), c_oracle.pipeline12_reinhard(packed[k], work="f16", out="f16"), f"frame {k}")
    # 70 small frames = two launches (64 + 6); 130 rows: the last band of every column of waves is partial
    small = [natural_packed12(np.random.default_rng(100 + k), 130, 1544) for k in range(3)]
    sm = [torch.from_numpy(p).to(dev) for p in small]
    ref = [pipeline12_reinhard(f, whole_frame=True).clone() for f in sm]
    bp = BatchPipeline(70, 130, 1544, dev, whole_frame=True, dtype=ti.types.u8, gamma=0.7)
    ref8 = [pipeline12_reinhard(f, whole_frame=True, dtype=ti.types.u8, gamma=0.7).clone() for f in sm]
    outs = bp([sm[i % 3] for i in range(70)])
    assert bp.check() == []
    for i in range(70):
        assert torch.equal(outs[i], ref8[i % 3]), f"small batch frame {i}"
    assert_close(ref[0].cpu().numpy(), c_oracle.pipeline12_reinhard(small[0], work="f16", out="f16"), "small frame")


def test_default_chain_is_the_whole_frame_kernel_when_it_fits(ti, dev, rng):
    """pipeline12_reinhard / BatchPipeline pick the single-launch kernel by themselves for frames it takes (and check its
    fault word), the multi-pass chain for the others."""
    from taichi_image_amd.pipeline import pipeline12_reinhard, BatchPipeline, whole_frame_fits
    packed = natural_packed12(rng, 96, 1024)
    f = torch.from_numpy(packed).to(dev)
    assert torch.equal(pipeline12_reinhard(f), pipeline12_reinhard(f, whole_frame=True))
    assert BatchPipeline(2, 96, 1024, dev).whole_frame
    # f32 work dtype, f32 output, the IDS layout: not the whole-frame kernel's - the multi-pass chain, same call
    assert not BatchPipeline(2, 96, 1024, dev, dtype=ti.types.f32).whole_frame
    got = pipeline12_reinhard(f, work_dtype=ti.types.f32, dtype=ti.types.f32)
    assert_close(got.cpu().numpy(), O.pipeline12_reinhard(packed, 0, False, None, "f32", "f32"), "auto -> multi-pass f32")
    ids = natural_packed12(rng, 96, 1024, ids_format=True)
    got = pipeline12_reinhard(torch.from_numpy(ids).to(dev), ids_format=True)
    assert_close(got.cpu().numpy(), O.pipeline12_reinhard(ids, 0, True), "auto -> multi-pass ids")
    assert not whole_frame_fits(3084, 4096)
    big = natural_packed12(rng, 3084, 4096)
    got = pipeline12_reinhard(torch.from_numpy(big).to(dev))
    assert_close(got.cpu().numpy(), c_oracle.pipeline12_reinhard(big, work="f16", out="f16"), "auto -> multi-pass big")


def test_whole_frame_timeout_is_reported_and_repaired(ti, dev, scenes):
    """A barrier that times out (provoked with a poll budget of one round) must not pass silently: the fault word of the
    frame and the device's mailbox are set, the checked call re-issues the frame through the multi-pass chain - the output
    then equals the oracle - and warns or raises; BatchPipeline.check() repairs a batch the same way."""
    from taichi_image_amd import _native
    from taichi_image_amd.pipeline import pipeline12_reinhard, BatchPipeline, WholeFrameTimeout
    L = _native.lib()
    packed = [packed_from(scenes[0]), packed_from(scenes[1], 0.7, 0.1)]
    frames = [torch.from_numpy(p).to(dev) for p in packed]
    refs = [c_oracle.pipeline12_reinhard(p, work="f16", out="f16") for p in packed]
    good = pipeline12_reinhard(frames[0], whole_frame=True).clone()
    torch.cuda.synchronize()
    assert L.mi_isp_whole_frame_faults(1) == 0
    L.mi_isp_whole_frame_set_poll_limit(1)
    try:
        out = torch.zeros_like(good)
        with pytest.raises(WholeFrameTimeout) as ei:
            pipeline12_reinhard(frames[0], out=out, whole_frame=True, check=True, on_timeout="raise")
        assert ei.value.frames == (0,)
        assert_close(out.cpu().numpy(), refs[0], "re-issued frame (raise)")
        # the default (auto-selected kernel): NO synchronisation; the fault is found by the FOLLOWING call's look at the
        # device's mailbox (a host read), which re-issues the lost frame and reports it
        from taichi_image_amd.pipeline import check_pending
        got = pipeline12_reinhard(frames[1])
        torch.cuda.synchronize()                                  # (so that the kernel has run and stored to the mailbox)
        with pytest.warns(RuntimeWarning, match="timed out"):
            got2 = pipeline12_reinhard(frames[0])                 # repairs `got`; its own launch is lost in turn ...
        torch.cuda.synchronize()
        with pytest.raises(WholeFrameTimeout):
            check_pending(dev, on_timeout="raise")                # ... and repaired here, on demand
        assert check_pending(dev) == 0                            # nothing is left pending
        assert_close(got.cpu().numpy(), refs[1], "re-issued frame (default, repaired by the following call)")
        assert_close(got2.cpu().numpy(), refs[0], "re-issued frame (default, repaired by check_pending)")
        # unchecked: the call returns, the mailbox tells without a synchronisation of the caller's
        pipeline12_reinhard(frames[0], out=out, whole_frame=True)
        torch.cuda.synchronize()
        assert L.mi_isp_whole_frame_faults(0) != 0 and _error_word(ti, 3072, 4096, dev) != 0
        assert _native.workspace(3072, 4096, dev) is not None
        bp = BatchPipeline(3, 3072, 4096, dev, whole_frame=True)
        outs = bp([frames[0], frames[1], frames[0]])
        with pytest.warns(RuntimeWarning, match="timed out"):
            lost = bp.check()
        assert len(lost) >= 1
        for o, k in zip(outs, (0, 1, 0)):
            assert_close(o.cpu().numpy(), refs[k], "repaired batch frame")
    finally:
        L.mi_isp_whole_frame_set_poll_limit(0)
        torch.cuda.synchronize()
        L.mi_isp_whole_frame_faults(1)
        # the fault word of the shared workspace is sticky: clear it for the tests that follow
        ws = _native.workspace(3072, 4096, dev)
        off = int(L.mi_isp_workspace_error_offset(3072, 4096))
        ws[off:off + 4].zero_()
    again = pipeline12_reinhard(frames[0], whole_frame=True)
    torch.cuda.synchronize()
    assert _error_word(ti, 3072, 4096, dev) == 0 and torch.equal(again, good)
    # the default path once more, undisturbed: asynchronous, nothing to report at the next look
    again = pipeline12_reinhard(frames[0])
    torch.cuda.synchronize()
    from taichi_image_amd.pipeline import check_pending
    assert check_pending(dev) == 0 and torch.equal(again, good)


def test_whole_frame_launch_that_lost_a_block_ends_soon(ti, dev, rng):
    """One block that never posts (what a block that is not resident looks like to the others) must cost ONE poll budget
    (~0.1 s), not one per barrier that is left: a 64-frame launch has 192 of them (round 3: ~19 s of a spinning resident
    grid).  Every lost frame carries its fault word and check() repairs the batch."""
    import time
    from taichi_image_amd import _native
    from taichi_image_amd.pipeline import BatchPipeline
    L = _native.lib()
    H, W = 96, 1024                                                # 4 blocks
    packed = [natural_packed12(np.random.default_rng(300 + k), H, W) for k in range(3)]
    frames = [torch.from_numpy(packed[k % 3]).to(dev) for k in range(64)]
    refs = [c_oracle.pipeline12_reinhard(p, work="f16", out="f16") for p in packed]
    bp = BatchPipeline(64, H, W, dev, whole_frame=True)
    outs = bp(frames)
    assert bp.check() == [] and not bp.faulted()
    for k in (0, 1, 2, 63):
        assert_close(outs[k].cpu().numpy(), refs[k % 3], f"undisturbed frame {k}")
    L.mi_isp_whole_frame_set_sabotage(1)
    try:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = bp(frames)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        L.mi_isp_whole_frame_set_sabotage(-1)
    assert bp.faulted(), "the mailbox must tell without a synchronisation of the library's"
    assert dt < 2.0, f"a launch that lost one block took {dt:.2f} s"
    with pytest.warns(RuntimeWarning, match="timed out"):
        lost = bp.check()
    assert 0 in lost
    for k in range(64):
        assert_close(outs[k].cpu().numpy(), refs[k % 3], f"repaired frame {k}")
    outs = bp(frames)                                              # and the next launch is undisturbed again
    assert bp.check() == []
    assert_close(outs[5].cpu().numpy(), refs[2], "frame after the repair")


@pytest.mark.parametrize("out", ["f16", "u8"])
def test_config3b_per_axis_resize_at_full_size(ti, dev, scenes, out):
    """BASELINE config 3's nominal 1920 x 1080: not reachable through Camera16 (uniform scale only) but
    through the resize primitive's per-axis scale (interpolate.py:71-86,128-139) = (1080 / 3072, 1920 / 4096) on the
    demosaiced 4K f16 image - bit-exact against the oracle at full size."""
    from taichi_image_amd import interpolate
    packed = packed_from(scenes[3])
    isp = ti.Camera16(ti.BayerPattern.RGGB, device=dev)
    rgb = isp.load_packed12(torch.from_numpy(packed).to(dev))
    ref_rgb = O.isp_load_packed12(packed, "f16")
    assert np.array_equal(rgb.cpu().numpy().view(np.uint16), ref_rgb.view(np.uint16))
    scale = (1080 / 3072, 1920 / 4096)
    assert scale == (0.3515625, 0.46875)
    got = interpolate.resize_bilinear(rgb, (1920, 1080), scale=scale, dtype=getattr(ti.types, out))
    ref = O.resize_bilinear(ref_rgb, (1920, 1080), scale=scale, dtype=out)
    assert got.shape == (1080, 1920, 3)
    assert np.array_equal(got.cpu().numpy().view(np.uint8), ref.view(np.uint8)), "per-axis resize at 4K is not bit-exact"


# ---- boundary: one C call per camera group (mi_isp_camera_frame_batch), straight through ctypes ------------------
@pytest.mark.parametrize("tonemap,resize_width", [(0, 0), (0, 960), (1, 960)])
def test_camera_frame_batch_through_the_c_abi(ti, dev, rng, tonemap, resize_width):
    """packed bytes of 3 cameras -> u8 outputs with ONE library call per group, over 3 groups (rolling metering):
    equal to the Python ISP (load_packed12 x n + tonemap_reinhard / tonemap_linear), which the other tests pin."""
    import ctypes
    from taichi_image_amd import _native
    L = _native.lib()
    H, W, n = 1536, 2048, 3
    isp = ti.Camera16(ti.BayerPattern.RGGB, moving_alpha=0.25, resize_width=resize_width, device=dev,
                      transform=ti.ImageTransform.rotate_90)
    Wd, Hd = (resize_width, round(H * resize_width / W)) if resize_width else (W, H)
    scale = resize_width / W if resize_width else 0.0
    state = torch.zeros(9, dtype=torch.float32, device=dev)
    images = [torch.empty((Hd, Wd, 3), dtype=torch.float16, device=dev) for _ in range(n)]
    outs = [torch.empty((Wd, Hd, 3), dtype=torch.uint8, device=dev) for _ in range(n)]      # rotate_90 swaps the dims
    ws = torch.zeros(int(L.mi_isp_workspace_bytes(Hd, Wd)), dtype=torch.uint8, device=dev)
    for group in range(3):
        packs = [torch.from_numpy(natural_packed12(rng, H, W, dark=0.03 * k + 0.02 * group)).to(dev) for k in range(n)]
        want_imgs = [isp.load_packed12(p) for p in packs]
        want = (isp.tonemap_reinhard(want_imgs, gamma=0.6, intensity=1.2, light_adapt=0.8, color_adapt=0.1) if tonemap == 0
                else isp.tonemap_linear(want_imgs, gamma=0.6))
        alpha = 0.0 if group == 0 else 1.0 - 0.25
        rc = L.mi_isp_camera_frame_batch(_native.ptr_array(packs), _native.ptr_array(images), _native.ptr_array(outs), n, H, W,
                                         12, 0, 0, None, ti.types.f16.code, Hd, Wd, ctypes.c_float(scale), 8,
                                         state.data_ptr(), ctypes.c_float(alpha), tonemap, ctypes.c_float(0.6),
                                         ctypes.c_float(1.2), ctypes.c_float(0.8), ctypes.c_float(0.1), 1,
                                         ws.data_ptr(), _native.stream_ptr(dev))
        assert rc == 0, L.mi_isp_last_error()
        torch.cuda.synchronize()
        assert torch.equal(state, isp.metrics), f"group {group}: metering state"
        for k in range(n):
            assert torch.equal(outs[k], want[k]), f"group {group} camera {k}: u8 output"
            assert torch.equal(images[k], want_imgs[k]), f"group {group} camera {k}: image left behind"


def test_camera_group_reinhard_through_the_c_abi_full_size(ti, dev, scenes):
    """mi_isp_camera_group_reinhard straight through ctypes at 4096 x 3072: three cameras, two groups (rolling metering),
    once without images (the reference bench's form) and once with: u8 outputs, p and state equal to the Python ISP's
    load_packed12 x n + tonemap_reinhard - which the other tests pin against the oracle - bit for bit."""
    import ctypes
    from taichi_image_amd import _native
    L = _native.lib()
    H, W, n = 3072, 4096, 3
    assert L.mi_isp_camera_group_fits(H, W, 0, ti.types.f16.code, 8) == 1
    isp = ti.Camera16(ti.BayerPattern.RGGB, moving_alpha=0.25, device=dev)
    state = torch.zeros(9, dtype=torch.float32, device=dev)
    images = [torch.empty((H, W, 3), dtype=torch.float16, device=dev) for _ in range(n)]
    outs = [torch.empty((H, W, 3), dtype=torch.uint8, device=dev) for _ in range(n)]
    ws = torch.zeros(int(L.mi_isp_workspace_bytes(H, W)) * (n + 1), dtype=torch.uint8, device=dev)
    scratch = torch.empty(int(L.mi_isp_camera_group_scratch_bytes(n, H, W)), dtype=torch.uint8, device=dev)
    for group in range(2):
        packs = [torch.from_numpy(packed_from(scenes[k], GAINS[(k + group) % len(GAINS)], OFFSETS[(k + group) % len(OFFSETS)])).to(dev)
                 for k in range(n)]
        want_imgs = [isp.load_packed12(p) for p in packs]
        want = isp.tonemap_reinhard(want_imgs, gamma=0.6, intensity=1.2, light_adapt=0.8, color_adapt=0.0)
        alpha = 0.0 if group == 0 else 1.0 - 0.25
        rc = L.mi_isp_camera_group_reinhard(_native.ptr_array(packs), _native.ptr_array(images) if group == 1 else None,
                                            _native.ptr_array(outs), n, H, W, 0, None, state.data_ptr(), state.data_ptr(), ctypes.c_float(alpha),
                                            ctypes.c_float(0.6), ctypes.c_float(1.2), ctypes.c_float(0.8), ctypes.c_float(0.0),
                                            scratch.data_ptr(), ws.data_ptr(), _native.stream_ptr(dev))
        assert rc == 0, L.mi_isp_last_error()
        torch.cuda.synchronize()
        assert L.mi_isp_camera_group_faults(0) == 0
        assert torch.equal(state, isp.metrics), f"group {group}: metering state"
        for k in range(n):
            assert torch.equal(outs[k], want[k]), f"group {group} camera {k}: u8 output"
            if group == 1:
                assert torch.equal(images[k].view(torch.int16), want_imgs[k].view(torch.int16)), f"group {group} camera {k}: p"


def test_isp_reuse_of_tonemapped_images_full_size(ti, dev, scenes):
    """Two 4096 x 3072 cameras through tonemap_reinhard twice (camera_isp.py:211 then :376-403 again): the second call
    meters the images the first one overwrote, not the subsample their load kernel left (the round-3 hole)."""
    from tests.util import reuse_case
    frames = [torch.from_numpy(packed_from(scenes[k], GAINS[k], OFFSETS[k])).to(dev) for k in range(2)]
    reuse_case(ti, dev, "Camera16", frames, "reinhard_twice")


def test_default_call_inside_a_stream_capture_takes_the_multi_pass_chain(ti, dev, rng):
    """the automatic choice must not put a whole-frame launch (which the library can neither order against other
    resident grids nor check) into a caller's capture.  Captured with torch.cuda.graph, replayed: the multi-pass chain's
    result, the mailbox untouched."""
    from taichi_image_amd import _native
    from taichi_image_amd.pipeline import pipeline12_reinhard
    packed = natural_packed12(rng, 96, 1024)
    f = torch.from_numpy(packed).to(dev)
    want = pipeline12_reinhard(f, whole_frame=False).clone()
    out = torch.empty_like(want)
    pipeline12_reinhard(f, out=out)                                 # warm (module load, workspace) outside the capture
    torch.cuda.synchronize()
    calls = []
    real = _native.lib().mi_isp_pipeline12_reinhard_whole_frame
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(device=dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        pipeline12_reinhard(f, out=out)                             # the side stream's own workspace, before the capture
        torch.cuda.synchronize()
        _native.lib().mi_isp_pipeline12_reinhard_whole_frame = lambda *a: calls.append(a) or real(*a)
        try:
            with torch.cuda.graph(g, stream=s):
                pipeline12_reinhard(f, out=out)
        finally:
            _native.lib().mi_isp_pipeline12_reinhard_whole_frame = real
    assert calls == [], "the whole-frame kernel was launched inside a capture"
    out.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want)
    assert _native.lib().mi_isp_whole_frame_faults(0) == 0


def test_isp_reinhard_4k_one_launch_equals_two_passes(ti, dev, scenes, monkeypatch):
    """The one-launch Reinhard on 4096 x 3072 f16 images (12 groups per thread: 7 kept in registers, 5 in LDS, one image at a
    time; opt-in - measured slower than the two passes): in-place p, u8 outputs and metrics equal those of the two passes
    bit for bit, three cameras."""
    frames = [torch.from_numpy(packed_from(scenes[k], GAINS[k], OFFSETS[k])).to(dev) for k in range(3)]

    def run(two_pass):
        monkeypatch.setenv("MI_ISP_REINHARD_LAUNCHES", "2" if two_pass else "1")
        isp = ti.Camera16(ti.BayerPattern.RGGB, moving_alpha=0.3, device=dev)
        imgs = [isp.load_packed12(f) for f in frames]
        outs = isp.tonemap_reinhard(imgs, gamma=0.6)
        torch.cuda.synchronize()
        return imgs, outs, isp.metrics.clone()
    i1, o1, m1 = run(False)
    i2, o2, m2 = run(True)
    assert torch.equal(m1, m2)
    for k in range(3):
        assert torch.equal(o1[k], o2[k]), f"image {k}: u8 output differs"
        assert torch.equal(i1[k].view(torch.int16), i2[k].view(torch.int16)), f"image {k}: in-place p differs"
    from taichi_image_amd import _native
    assert _native.lib().mi_isp_reinhard_faults(1) == 0
