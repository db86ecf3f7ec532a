"""Process a multi-camera scan of raw packed-12 frames: the reference's `scripts/tonemap_scan.py`
(:20-181) on the MI355X path.  Same arguments and flow - find the camera folders and their common image
names, load the raw bytes on a thread pool with one group of look-ahead, `Camera32.load_packed12` per
camera, one `tonemap_reinhard` over the group, concatenate the outputs as a grid, optionally write it.

Differences, all on the host side: no OpenCV / natsort / tqdm (not needed on the box): names are sorted
naturally with a regex key, the grid is written as PNG (zlib) instead of JPEG, nothing is displayed.

    python -m taichi_image_amd.scripts.tonemap_scan --scan /data/scan_01 --width 4096 --write out/
"""
from __future__ import annotations

import argparse
import re
import struct
import zlib
from functools import partial
from pathlib import Path
from typing import List, Tuple

import numpy as np
import torch


def natural_key(name: str):
    """'img10' after 'img9' (natsort.natsorted in the reference, scripts/tonemap_scan.py:23,48)."""
    return [int(tok) if tok.isdigit() else tok.lower() for tok in re.split(r"(\d+)", str(name))]


def is_image_file(f: Path) -> bool:
    return f.is_file() and f.suffix in [".tiff", ".raw"]            # :20-21


def find_images(folder: Path) -> List[str]:
    return sorted([f.name for f in folder.iterdir() if is_image_file(f)], key=natural_key)   # :23-24


def find_folder_images(folder: Path) -> Tuple[List[Path], List[str]]:
    return [folder], find_images(folder)                             # :27-28


def set_intersections(image_sets):
    common = set(image_sets[0])                                      # :32-36
    for images in image_sets[1:]:
        common.intersection_update(set(images))
    return list(common)


def find_scan_images(scan_folder: Path) -> Tuple[List[Path], List[str]]:
    """:39-53: camera folders (those holding images) and the image names common to all of them."""
    cam_folders = {}
    for f in scan_folder.iterdir():
        if f.is_dir():
            images = find_images(f)
            if len(images) > 0:
                cam_folders[f.name] = images
    if not cam_folders:
        raise ValueError(f"No image folders found in {scan_folder}")
    common_images = set_intersections(list(cam_folders.values()))
    cam_ids = sorted(cam_folders.keys(), key=natural_key)
    if len(common_images) == 0:
        raise ValueError(f"No common images found in {cam_ids}")
    print(f"Found {cam_ids} image folders with {len(common_images)} images")
    return [scan_folder / cid for cid in cam_ids], sorted(common_images, key=natural_key)


def find_scan_folders(scan_folder: Path) -> Tuple[List[Path], List[str]]:
    folder = Path(scan_folder)                                       # :56-61
    if not folder.is_dir():
        raise FileNotFoundError(f"Folder {folder} does not exist or is not a directory")
    return find_scan_images(folder)


def concat_image_grid(images: list, rows: int) -> torch.Tensor:
    """:90-101: rows of horizontally concatenated images, concatenated vertically."""
    n_images = len(images)
    n_cols = (n_images + rows - 1) // rows
    grid_rows = []
    for i in range(0, n_images, n_cols):
        grid_rows.append(torch.concat(images[i:i + n_cols], dim=1))
    return torch.concat(grid_rows, dim=0)


def write_png(path, image: np.ndarray) -> None:
    """8-bit RGB PNG with the standard library only (the reference writes JPEG through OpenCV, :176-178)."""
    assert image.ndim == 3 and image.shape[2] == 3 and image.dtype == np.uint8
    h, w, _ = image.shape
    raw = np.concatenate([np.zeros((h, 1), np.uint8), image.reshape(h, w * 3)], axis=1).tobytes()   # filter 0 per row

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)))
        f.write(chunk(b"IDAT", zlib.compress(raw, 3)))
        f.write(chunk(b"IEND", b""))


def build_parser() -> argparse.ArgumentParser:
    from ..interpolate import ImageTransform
    parser = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    parser.add_argument("--scan", type=Path)
    parser.add_argument("--images", type=Path)
    parser.add_argument("--reverse", action="store_true")
    parser.add_argument("--width", type=int, default=4096)
    # tonemap parameters (defaults of the reference, :115-124)
    parser.add_argument("--gamma", type=float, default=0.9)
    parser.add_argument("--intensity", type=float, default=3.0)
    parser.add_argument("--color_adapt", type=float, default=0.0)
    parser.add_argument("--light_adapt", type=float, default=0.9)
    parser.add_argument("--moving_alpha", type=float, default=0.02)
    parser.add_argument("--resize_width", type=int, default=0)
    parser.add_argument("--transform", type=lambda s: ImageTransform[s] if s in ImageTransform.__members__ else ImageTransform(s),
                        default=ImageTransform.rotate_90)
    parser.add_argument("--correct_colors", action="store_true")
    parser.add_argument("--write", type=Path, default=None)
    parser.add_argument("--rows", type=int, default=2)
    parser.add_argument("--device", default="cuda:0")
    parser.add_argument("--ids_format", action="store_true")
    return parser


def main(argv=None) -> int:
    from .. import bayer, camera_isp, ingest
    args = build_parser().parse_args(argv)
    device = torch.device(args.device)
    isp = camera_isp.Camera32(bayer.BayerPattern.RGGB, transform=args.transform, moving_alpha=args.moving_alpha,
                              resize_width=args.resize_width, correct_colors=args.correct_colors, device=device)
    if args.scan is not None:
        folders, names = find_scan_folders(args.scan)
    elif args.images is not None:
        folders, names = find_folder_images(args.images)
    else:
        raise ValueError("No --scan or --images specified")
    if args.reverse:
        names = list(reversed(names))
    groups = ingest.load_images_iter(partial(ingest.load_raw_bytes, device=device), folders, names)

    def load_image(raw: torch.Tensor) -> torch.Tensor:
        assert raw.shape[0] % 2 == 0, "bytes must have an even number"                 # :158
        return isp.load_packed12(raw.view(-1, (args.width * 3) // 2), ids_format=args.ids_format)

    n = 0
    for name, group in groups:
        images = [load_image(raw) for raw in group.values()]
        outputs = isp.tonemap_reinhard(images, gamma=args.gamma, intensity=args.intensity,
                                       color_adapt=args.color_adapt, light_adapt=args.light_adapt)
        image = concat_image_grid(outputs, rows=args.rows).cpu().numpy()
        if args.write is not None:
            args.write.mkdir(exist_ok=True, parents=True)
            filename = args.write / f"{Path(name).stem}.png"
            print(f"Writing {filename}")
            write_png(filename, image)
        n += 1
    print(f"processed {n} image groups from {len(folders)} cameras")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
