"""The scan CLI (scripts/tonemap_scan.py of the reference): host logic on CPU, end to end on the GPU."""
import struct
import zlib

import numpy as np
import pytest

from taichi_image_amd.scripts import tonemap_scan as ts


def _read_png(path):
    data = open(path, "rb").read()
    assert data[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, shape = 8, b"", None
    while pos < len(data):
        n, tag = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(tag + body) & 0xFFFFFFFF
        if tag == b"IHDR":
            w, h, depth, ctype = struct.unpack(">IIBB", body[:10])
            assert (depth, ctype) == (8, 2)
            shape = (h, w)
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(shape[0], shape[1] * 3 + 1)
    assert (raw[:, 0] == 0).all()
    return raw[:, 1:].reshape(shape[0], shape[1], 3)


def test_natural_sort_and_discovery(tmp_path):
    assert sorted(["img10.raw", "img9.raw", "IMG1.raw"], key=ts.natural_key) == ["IMG1.raw", "img9.raw", "img10.raw"]
    for cam, names in (("cam2", ["f1.raw", "f2.raw", "f10.raw"]), ("cam10", ["f1.raw", "f10.raw", "skip.txt"]), ("empty", [])):
        (tmp_path / cam).mkdir()
        for n in names:
            (tmp_path / cam / n).write_bytes(b"\0\0")
    index = ts.ScanIndex.of_scan(tmp_path)
    assert [f.name for f in index.cameras] == ["cam2", "cam10"] and index.frames == ["f1.raw", "f10.raw"]
    single = ts.ScanIndex.of_directory(tmp_path / "cam2")
    assert single.cameras == [tmp_path / "cam2"] and single.frames == ["f1.raw", "f2.raw", "f10.raw"]
    with pytest.raises(FileNotFoundError):
        ts.ScanIndex.of_scan(tmp_path / "missing")
    (tmp_path / "cam3").mkdir(); (tmp_path / "cam3" / "other.raw").write_bytes(b"\0\0")
    with pytest.raises(ValueError):
        ts.ScanIndex.of_scan(tmp_path)
    (tmp_path / "void").mkdir()
    with pytest.raises(ValueError):
        ts.ScanIndex.of_scan(tmp_path / "void")


def test_png_writer_round_trip(tmp_path):
    img = np.random.default_rng(1).integers(0, 256, (7, 5, 3)).astype(np.uint8)
    ts.write_png(tmp_path / "x.png", img)
    assert np.array_equal(_read_png(tmp_path / "x.png"), img)


def test_grid_and_parser():
    import torch
    ims = [torch.full((2, 3, 3), i, dtype=torch.uint8) for i in range(5)]
    g = ts.tile_grid(ims[:4], rows=2)
    assert g.shape == (4, 6, 3) and int(g[0, 3, 0]) == 1 and int(g[2, 0, 0]) == 2
    a = ts.build_parser().parse_args(["--images", "x", "--transform", "none", "--rows", "1"])
    assert a.transform.value == "none" and a.gamma == 0.9 and a.intensity == 3.0 and a.moving_alpha == 0.02


@pytest.mark.gpu
def test_scan_end_to_end(tmp_path):
    """Two cameras x three frames of synthetic packed-12 raw files -> PNG grids; the first grid equals the
    library called directly."""
    import torch
    import taichi_image_amd as ti
    from taichi_image_amd import synthetic
    H, W = 64, 128
    frames = {}
    for c, cam in enumerate(("cam0", "cam1")):
        (tmp_path / "scan" / cam).mkdir(parents=True)
        for k in range(3):
            frames[(cam, k)] = synthetic.synthetic_packed12(3 * c + k, H, W)
            (tmp_path / "scan" / cam / f"frame{k}.raw").write_bytes(frames[(cam, k)].tobytes())
    out = tmp_path / "out"
    rc = ts.main(["--scan", str(tmp_path / "scan"), "--width", str(W), "--write", str(out), "--rows", "1",
                  "--transform", "none", "--moving_alpha", "0.1"])
    assert rc == 0
    pngs = sorted(p.name for p in out.iterdir())
    assert pngs == ["frame0.png", "frame1.png", "frame2.png"]
    got = _read_png(out / "frame0.png")
    assert got.shape == (H, 2 * W, 3)
    dev = torch.device("cuda", 0)
    isp = ti.Camera32(ti.BayerPattern.RGGB, moving_alpha=0.1, device=dev)
    imgs = [isp.load_packed12(torch.from_numpy(frames[(cam, 0)]).to(dev)) for cam in ("cam0", "cam1")]
    want = isp.tonemap_reinhard(imgs, gamma=0.9, intensity=3.0, color_adapt=0.0, light_adapt=0.9)
    assert np.array_equal(got, torch.concat(want, dim=1).cpu().numpy())
