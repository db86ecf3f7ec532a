"""Synthetic RAW frames of the benchmark workload; host-side numpy, input
generation only -- not part of the device path."""
from __future__ import annotations

import numpy as np


def synthetic_scene(k: int, H: int = 3072, W: int = 4096) -> np.ndarray:
    """Smooth-plus-noise RGB scene k in [0,1], float32 (H, W, 3)."""
    rng = np.random.default_rng(1234 + k)
    r = np.arange(H, dtype=np.float64)[:, None]
    c = np.arange(W, dtype=np.float64)[None, :]
    base = 0.05 + 0.9 * (0.5 + 0.5 * np.sin(2 * np.pi * (3 * r / H + k / 64))) * (0.5 + 0.5 * np.cos(2 * np.pi * 5 * c / W))
    img = np.empty((H, W, 3), dtype=np.float32)
    for ch, gain in enumerate((1.0, 0.8, 0.6)):
        img[..., ch] = np.clip(base * gain + rng.normal(0, 0.02, size=(H, W)), 0, 1)
    return img


def mosaic_rggb(img: np.ndarray) -> np.ndarray:
    """RGGB colour filter array of an RGB image (R at even/even, B at odd/odd)."""
    cfa = np.empty(img.shape[:2], dtype=img.dtype)
    cfa[0::2, 0::2] = img[0::2, 0::2, 0]
    cfa[0::2, 1::2] = img[0::2, 1::2, 1]
    cfa[1::2, 0::2] = img[1::2, 0::2, 1]
    cfa[1::2, 1::2] = img[1::2, 1::2, 2]
    return cfa


def pack12(v12: np.ndarray) -> np.ndarray:
    """Standard 12-bit packing, little-endian bit order: 2 px -> 3 bytes; (H, W) -> (H, W*3/2)."""
    p0 = v12[..., 0::2].astype(np.uint32)
    p1 = v12[..., 1::2].astype(np.uint32)
    out = np.empty(v12.shape[:-1] + (v12.shape[-1] // 2, 3), dtype=np.uint8)
    out[..., 0] = p0 & 0xFF
    out[..., 1] = ((p1 & 0xF) << 4) | (p0 >> 8)
    out[..., 2] = p1 >> 4
    return out.reshape(v12.shape[:-1] + (v12.shape[-1] * 3 // 2,))


def synthetic_packed12(k: int, H: int = 3072, W: int = 4096) -> np.ndarray:
    """Frame k: RGGB mosaic of scene k, quantised to 12 bit, standard packing -> u8 (H, W*3/2)."""
    cfa = mosaic_rggb(synthetic_scene(k, H, W))
    v12 = np.rint(cfa.astype(np.float64) * 4095).astype(np.uint16)
    return pack12(v12)
