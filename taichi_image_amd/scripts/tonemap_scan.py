"""Tonemap a multi-camera scan of raw packed-12 frames (the command line of the reference's
`scripts/tonemap_scan.py:104-128`, same arguments and defaults) on the MI355X path.

    python -m taichi_image_amd.scripts.tonemap_scan --scan /data/scan_01 --width 4096 --write out/

A scan is a directory of camera directories holding `.raw` / `.tiff` frames; the frames present in EVERY camera are
processed in natural name order: per frame one `Camera32.load_packed12` per camera and one `tonemap_reinhard` over the
camera group (so the rolling metering is shared, camera_isp.py:376-403), the u8 outputs tiled into a grid.
Host side without OpenCV / natsort / tqdm: grids are written as PNG, nothing is displayed.
"""
from __future__ import annotations

import argparse
import re
import struct
import zlib
from dataclasses import dataclass, field
from pathlib import Path
from typing import Dict, Iterator, List, Sequence

import numpy as np
import torch

RAW_SUFFIXES = (".raw", ".tiff")
_DIGITS = re.compile(r"(\d+)")


def natural_key(name) -> list:
    """Sort key under which 'img10' follows 'img9' (digit runs compare as numbers, the rest case-insensitively)."""
    return [(0, int(tok)) if tok.isdigit() else (1, tok.lower()) for tok in _DIGITS.split(str(name)) if tok != ""]


@dataclass
class ScanIndex:
    """Which frames to process and where each camera keeps them."""
    cameras: List[Path]                                   # camera directories, natural order
    frames: List[str] = field(default_factory=list)       # file names present in every camera, natural order

    @staticmethod
    def _raw_names(directory: Path) -> set:
        return {e.name for e in directory.iterdir() if e.is_file() and e.suffix in RAW_SUFFIXES}

    @classmethod
    def of_directory(cls, directory) -> "ScanIndex":
        """A single camera: every raw frame of one directory (--images)."""
        directory = Path(directory)
        if not directory.is_dir():
            raise FileNotFoundError(f"Folder {directory} does not exist or is not a directory")
        return cls([directory], sorted(cls._raw_names(directory), key=natural_key))

    @classmethod
    def of_scan(cls, scan_dir) -> "ScanIndex":
        """Every sub-directory with raw frames is a camera; only frames that all cameras have are kept (--scan)."""
        scan_dir = Path(scan_dir)
        if not scan_dir.is_dir():
            raise FileNotFoundError(f"Folder {scan_dir} does not exist or is not a directory")
        per_camera: Dict[Path, set] = {}
        for sub in scan_dir.iterdir():
            if sub.is_dir():
                names = cls._raw_names(sub)
                if names:
                    per_camera[sub] = names
        if not per_camera:
            raise ValueError(f"No image folders found in {scan_dir}")
        shared = set.intersection(*per_camera.values())
        cameras = sorted(per_camera, key=lambda d: natural_key(d.name))
        if not shared:
            raise ValueError(f"No common images found in {[c.name for c in cameras]}")
        return cls(cameras, sorted(shared, key=natural_key))

    def groups(self, loader, reverse: bool = False) -> Iterator:
        """(frame name, [one loaded frame per camera]) with the next group's files already loading (ingest)."""
        from .. import ingest
        names = list(reversed(self.frames)) if reverse else self.frames
        if not names:
            return
        for name, by_camera in ingest.load_images_iter(loader, self.cameras, names):
            yield name, [by_camera[c] for c in self.cameras]


def tile_grid(images: Sequence[torch.Tensor], rows: int) -> torch.Tensor:
    """Images of equal size tiled row-major into `rows` rows (the last row may be shorter only if it is the only one)."""
    per_row = -(-len(images) // max(1, rows))
    strips = [torch.cat(list(images[i:i + per_row]), dim=1) for i in range(0, len(images), per_row)]
    return torch.cat(strips, dim=0)


def write_png(path, image: np.ndarray) -> None:
    """8-bit RGB PNG with the standard library only (the reference writes JPEG through OpenCV)."""
    assert image.ndim == 3 and image.shape[2] == 3 and image.dtype == np.uint8
    h, w, _ = image.shape
    scanlines = np.concatenate([np.zeros((h, 1), np.uint8), image.reshape(h, w * 3)], axis=1).tobytes()   # filter type 0

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(scanlines, 3)) + chunk(b"IEND", b""))


def build_parser() -> argparse.ArgumentParser:
    """The argument surface of scripts/tonemap_scan.py:104-128 (+ --device, --ids_format)."""
    from ..interpolate import ImageTransform

    def transform(s):
        return ImageTransform[s] if s in ImageTransform.__members__ else ImageTransform(s)

    ap = argparse.ArgumentParser(description="Tonemap a scan of raw packed-12 camera frames on an MI355X")
    src = ap.add_argument_group("input")
    src.add_argument("--scan", type=Path)
    src.add_argument("--images", type=Path)
    src.add_argument("--reverse", action="store_true")
    src.add_argument("--width", type=int, default=4096)
    src.add_argument("--ids_format", action="store_true")
    tone = ap.add_argument_group("tonemap")
    tone.add_argument("--gamma", type=float, default=0.9)
    tone.add_argument("--intensity", type=float, default=3.0)
    tone.add_argument("--color_adapt", type=float, default=0.0)
    tone.add_argument("--light_adapt", type=float, default=0.9)
    tone.add_argument("--moving_alpha", type=float, default=0.02)
    tone.add_argument("--resize_width", type=int, default=0)
    tone.add_argument("--transform", type=transform, default=ImageTransform.rotate_90)
    tone.add_argument("--correct_colors", action="store_true")
    out = ap.add_argument_group("output")
    out.add_argument("--write", type=Path, default=None)
    out.add_argument("--rows", type=int, default=2)
    out.add_argument("--device", default="cuda:0")
    return ap


def main(argv=None) -> int:
    from functools import partial
    from .. import bayer, camera_isp, ingest
    args = build_parser().parse_args(argv)
    if args.scan is None and args.images is None:
        raise ValueError("No --scan or --images specified")
    index = ScanIndex.of_scan(args.scan) if args.scan is not None else ScanIndex.of_directory(args.images)
    print(f"{len(index.cameras)} camera(s) {[c.name for c in index.cameras]}, {len(index.frames)} frame(s) each")
    device = torch.device(args.device)
    isp = camera_isp.Camera32(bayer.BayerPattern.RGGB, transform=args.transform, moving_alpha=args.moving_alpha,
                              resize_width=args.resize_width, correct_colors=args.correct_colors, device=device)
    row_bytes = args.width * 3 // 2
    if args.write is not None:
        args.write.mkdir(exist_ok=True, parents=True)
    done = 0
    for name, raws in index.groups(partial(ingest.load_raw_bytes, device=device), args.reverse):
        for raw in raws:
            assert raw.numel() % row_bytes == 0, f"{name}: {raw.numel()} bytes is not a whole number of {row_bytes}-byte rows"
        images = [isp.load_packed12(raw.view(-1, row_bytes), ids_format=args.ids_format) for raw in raws]
        outputs = isp.tonemap_reinhard(images, gamma=args.gamma, intensity=args.intensity,
                                       color_adapt=args.color_adapt, light_adapt=args.light_adapt)
        if args.write is not None:
            target = args.write / (Path(name).stem + ".png")
            write_png(target, tile_grid(outputs, args.rows).cpu().numpy())
            print(f"wrote {target}")
        done += 1
    print(f"processed {done} frame group(s) from {len(index.cameras)} camera(s)")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
