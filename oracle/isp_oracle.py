"""NumPy CPU restatement of the taichi_image camera-ISP hot path.

TEST INFRASTRUCTURE ONLY -- never imported by the product package.

PARITY STATUS: *parity unpinned* by the reference for everything except the packed
12-bit encode->decode round trip (the only assertion the reference ships:
taichi_image/test/packed.py:6-15).  The reference cannot be executed here (its
``taichi`` dependency is not installed and cannot be installed), and it ships no golden
vectors.  What pins each stage of this restatement instead (tests/test_oracle.py unless noted):

  stage (reference lines)                         pinned by
  ----------------------------------------------  ------------------------------------------------------------
  decode12 / encode12 (packed.py:24-55,92-131)    the reference's round trip; hand KATs of both bit layouts and of
                                                  the scaled f16 bit patterns (tests/golden/kat.json)
  weight tables, pattern maps (bayer.py:15-97)    KAT tables; each channel sums to 16; equality with the four
                                                  published Malvar-He-Cutler 2004 filters (interior, 4 patterns)
  border renormalisation (bayer.py:138-155)       rebuilt pixel by pixel from the published 5x5 filters on images in
                                                  which every pixel is a border pixel (4x4, 6x8, 2x6; 4 patterns)
  bilinear resize (interpolate.py:24-34,59-66)    hand-computed 3x3 -> 2x2 and clamp-to-edge cases; identity / edge
                                                  properties
  orientation transforms (interpolate.py:36-54)   group identities
  ISP Reinhard (camera_isp.py:186-218)            the published photoreceptor model (Reinhard & Devlin 2005, eqs. 1-7,
                                                  float64) on a 2x2 image with hand-set metering, three parameter sets
  stateless Reinhard (tonemap.py:108-131)         the same model incl. the (log_min, -log_max) quirk of tonemap.py:102
  metering (camera_isp.py:142-175)                double-blended bounds and shard-combination properties; the C
                                                  restatement over three steps (tests/test_c_oracle.py)
  everything                                      a second, independent implementation (oracle/isp_oracle.c)

Every function cites the reference lines it restates (paths relative to
/root/reference/taichi_image/).  Arithmetic is carried out in float32 in the same
operation order as the reference kernels; reductions (sums) are evaluated in float64
and rounded once, because the reference's own atomic-add order is non-deterministic.

Conventions: arrays are [row, col] / [row, col, ch]; dtypes are the strings
'u8', 'u16', 'i16', 'f16', 'f32' (types.py:12-18 scale factors).

Defined behaviour where the reference is undefined (fptoui of NaN / out-of-range):
NaN -> 0 and saturation to the integer range.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32

SCALE = {"u8": 255.0, "u16": 65535.0, "i16": 32767.0, "f16": 1.0, "f32": 1.0}  # types.py:12-18
NP_DTYPE = {"u8": np.uint8, "u16": np.uint16, "i16": np.int16, "f16": np.float16, "f32": np.float32}
_INT_RANGE = {"u8": (0, 255), "u16": (0, 65535), "i16": (-32768, 32767)}

GRAY_W = np.array([0.299, 0.587, 0.114], dtype=f32)  # color/__init__.py:7-10


def dtype_name(arr) -> str:
    return {np.dtype(v): k for k, v in NP_DTYPE.items()}[np.dtype(arr.dtype)]


def cast_out(x: np.ndarray, dtype: str) -> np.ndarray:
    """ti.cast(float32 -> dtype): RNE for floats, truncation toward zero for ints."""
    x = np.asarray(x, dtype=f32)
    if dtype in ("f16", "f32"):
        with np.errstate(over="ignore"):
            return x.astype(NP_DTYPE[dtype])
    lo, hi = _INT_RANGE[dtype]
    y = np.where(np.isnan(x), f32(0), x)
    y = np.clip(np.trunc(y), lo, hi)
    return y.astype(NP_DTYPE[dtype])


# --------------------------------------------------------------------------------------
# packed.py
# --------------------------------------------------------------------------------------

def decode12_pairs(enc: np.ndarray, ids_format: bool = False) -> np.ndarray:
    """3 bytes -> two 12-bit values.  packed.py:24-31 (standard), :37-44 (IDS)."""
    b = enc.reshape(-1, 3).astype(np.uint16)
    b0, b1, b2 = b[:, 0], b[:, 1], b[:, 2]
    if not ids_format:
        p0 = ((b1 & 0xF) << 8) | b0
        p1 = (b2 << 4) | (b1 >> 4)
    else:
        p0 = (b0 << 4) | (b2 & 0xF)
        p1 = (b1 << 4) | (b2 >> 4)
    return np.stack([p0, p1], axis=1).reshape(-1).astype(np.uint16)


def _write_scaled(v_u16: np.ndarray, dtype: str, denom: float) -> np.ndarray:
    # packed.py:98-100 / :139-141 -- multiply by a pre-rounded reciprocal constant
    k = f32(SCALE[dtype] / denom)
    return cast_out(v_u16.astype(f32) * k, dtype)


def decode12(values: np.ndarray, dtype: str = "u16", scaled: bool = False, ids_format: bool = False):
    """packed.py:188-198 (+ kernel :92-131)."""
    assert values.dtype == np.uint8
    shape = values.shape
    assert shape[-1] % 3 == 0
    v = decode12_pairs(np.ascontiguousarray(values).reshape(-1), ids_format)
    out = _write_scaled(v, dtype, 4095.0) if scaled else v.astype(NP_DTYPE[dtype])
    return out.reshape(shape[:-1] + (shape[-1] * 2 // 3,))


def decode16(values: np.ndarray, dtype: str = "u16", scaled: bool = False):
    """packed.py:149-157 (little-endian pairs); the module-level wrapper :200-210 is broken
    in the reference (passes an unknown kwarg) -- semantics taken from the kernel."""
    assert values.dtype == np.uint8
    shape = values.shape
    assert shape[-1] % 2 == 0
    b = np.ascontiguousarray(values).reshape(-1, 2).astype(np.uint16)
    v = (b[:, 1] << 8) | b[:, 0]
    out = _write_scaled(v, dtype, 65535.0) if scaled else v.astype(NP_DTYPE[dtype])
    return out.reshape(shape[:-1] + (shape[-1] // 2,))


def _round_half_away(x: np.ndarray) -> np.ndarray:
    # ti.round == llvm.round (half away from zero); exact for |x| < 2**22
    r = np.trunc(x)
    return r + np.where(np.abs(x - r) >= f32(0.5), np.sign(x), f32(0)).astype(f32)


def encode12(values: np.ndarray, scaled: bool = False, ids_format: bool = False) -> np.ndarray:
    """packed.py:176-185 (+ :13-20, :48-55, :60-89)."""
    shape = values.shape
    assert shape[-1] % 2 == 0
    flat = np.ascontiguousarray(values).reshape(-1)
    if scaled:
        k = f32(4095.0 / SCALE[dtype_name(values)])
        v = _round_half_away(flat.astype(f32) * k)
        v = np.clip(v, 0, 65535).astype(np.uint16)
    else:
        v = flat.astype(np.uint16)
    p0 = v[0::2].astype(np.uint32)
    p1 = v[1::2].astype(np.uint32)
    if not ids_format:
        e = np.stack([p0 & 0xFF, ((p1 & 0xF) << 4) | (p0 >> 8), p1 >> 4], axis=1)
    else:
        e = np.stack([p0 >> 4, p1 >> 4, ((p0 & 0xF) << 4) | (p1 & 0xF)], axis=1)
    e = (e & 0xFF).astype(np.uint8).reshape(-1)
    return e.reshape(shape[:-1] + (shape[-1] * 3 // 2,))


# --------------------------------------------------------------------------------------
# camera_isp.py loaders (K11)
# --------------------------------------------------------------------------------------

def load_16u(image: np.ndarray, dtype: str) -> np.ndarray:
    """camera_isp.py:82-87: cast(f32(u16)/65535.0, dtype) -- a true division."""
    return cast_out(image.astype(f32) / f32(65535.0), dtype)


def load_32f(image: np.ndarray, dtype: str) -> np.ndarray:
    """camera_isp.py:89-93."""
    return cast_out(image.astype(f32), dtype)


def load_16f(image: np.ndarray, dtype: str) -> np.ndarray:
    """camera_isp.py:95-99: the u16 input is converted numerically, not bit-cast."""
    return cast_out(image.astype(f32), dtype)


# --------------------------------------------------------------------------------------
# bayer.py
# --------------------------------------------------------------------------------------

RGGB, GRBG, GBRG, BGGR = 0, 1, 2, 3  # bayer.py:75-79

PIXEL_ORDER = {RGGB: (0, 1, 1, 2), GRBG: (1, 0, 2, 1), GBRG: (1, 2, 0, 1), BGGR: (2, 1, 1, 0)}  # :85-90
# kernels used at (even row, even col), (odd row, even col), (even row, odd col), (odd, odd) :92-97,165-175
KERNEL_PATTERN = {RGGB: (0, 1, 2, 3), GBRG: (1, 0, 3, 2), GRBG: (2, 3, 0, 1), BGGR: (3, 2, 1, 0)}

# 13-tap diamond in the reference's enumeration order (bayer.py:15-27), (d_row, d_col)
DIAMOND = [(-2, 0), (-1, -1), (-1, 0), (-1, 1), (0, -2), (0, -1), (0, 0), (0, 1), (0, 2),
           (1, -1), (1, 0), (1, 1), (2, 0)]


def _diamond_from_wedge(a, b, c):
    """Expand the upper-left wedge of a 4-fold symmetric 5x5 diamond into the 13 taps
    (kernel.py:3-12 'symmetrical' applied to the three partial rows of bayer.py:34-43)."""
    (a0,), (b0, b1), (c0, c1, c2) = a, b, c
    top, mid, ctr = [a0], [b0, b1, b0], [c0, c1, c2, c1, c0]
    return top + mid + ctr + mid + top


def bayer_kernels() -> np.ndarray:
    """[4 kernels][13 taps][3 channels] integer weights.  bayer.py:30-55."""
    g_rb = _diamond_from_wedge((-2,), (0, 4), (-2, 4, 8))     # G at R/B sites
    r_g1 = _diamond_from_wedge((-2,), (-2, 8), (1, 0, 10))    # R at G1, B at G2
    r_g2 = _diamond_from_wedge((1,), (-2, 0), (-2, 8, 10))    # B at G1, R at G2
    rb_br = _diamond_from_wedge((-3,), (4, 0), (-3, 0, 12))   # R at B, B at R
    ident = _diamond_from_wedge((0,), (0, 0), (0, 0, 16))
    b_g1, b_g2 = r_g2, r_g1
    per_site = [
        (ident, g_rb, rb_br),   # K0: red site
        (r_g1, ident, b_g1),    # K1: green site, red above/below
        (r_g2, ident, b_g2),    # K2: green site, red left/right
        (rb_br, g_rb, ident),   # K3: blue site
    ]
    k = np.zeros((4, 13, 3), dtype=np.int32)
    for i, chans in enumerate(per_site):
        for ch in range(3):
            k[i, :, ch] = chans[ch]
    return k


BAYER_KERNELS = bayer_kernels()


def rgb_to_bayer(image: np.ndarray, pattern: int = RGGB) -> np.ndarray:
    """bayer.py:101-112, 193-198."""
    assert image.ndim == 3 and image.shape[2] == 3
    p1, p2, p3, p4 = PIXEL_ORDER[pattern]
    h, w = image.shape[:2]
    h2, w2 = (h // 2) * 2, (w // 2) * 2
    out = np.zeros((h, w), dtype=image.dtype)
    out[0:h2:2, 0:w2:2] = image[0:h2:2, 0:w2:2, p1]
    out[0:h2:2, 1:w2:2] = image[0:h2:2, 1:w2:2, p2]
    out[1:h2:2, 0:w2:2] = image[1:h2:2, 0:w2:2, p3]
    out[1:h2:2, 1:w2:2] = image[1:h2:2, 1:w2:2, p4]
    return out


def bayer_to_rgb(cfa: np.ndarray, pattern: int = RGGB, correct_colors=None, dtype: str | None = None):
    """bayer.py:115-177 (filter_at :138-155, write_pixel :133-134), :202-219.

    Per pixel: c = sum_over_in_bounds_taps f32(cfa)*w (sequential, reference tap order),
    t = sum of in-bounds weights; c /= in_scale*t; optional M@c; clamp; cast(c*out_scale).
    """
    assert cfa.ndim == 2 and cfa.shape[0] % 2 == 0 and cfa.shape[1] % 2 == 0
    in_dtype = dtype_name(cfa)
    out_dtype = in_dtype if dtype is None else dtype
    in_scale, out_scale = f32(SCALE[in_dtype]), f32(SCALE[out_dtype])
    H, W = cfa.shape
    P = np.zeros((H + 4, W + 4), dtype=f32)
    P[2:-2, 2:-2] = cfa.astype(f32)
    V = np.zeros((H + 4, W + 4), dtype=f32)
    V[2:-2, 2:-2] = 1
    out = np.empty((H, W, 3), dtype=f32)
    kp = KERNEL_PATTERN[pattern]
    for site, (i, k) in enumerate([(0, 0), (1, 0), (0, 1), (1, 1)]):
        weights = BAYER_KERNELS[kp[site]]
        h2, w2 = H // 2, W // 2
        c = np.zeros((h2, w2, 3), dtype=f32)
        t = np.zeros((h2, w2, 3), dtype=f32)
        for (dr, dc), w3 in zip(DIAMOND, weights):
            if not np.any(w3):
                continue
            r0, c0 = 2 + i + dr, 2 + k + dc
            x = P[r0:r0 + H:2, c0:c0 + W:2]
            v = V[r0:r0 + H:2, c0:c0 + W:2]
            w3f = w3.astype(f32)
            c += x[..., None] * w3f          # f32 multiply then f32 add, as cfa*vec3(w) ; c +=
            t += v[..., None] * w3f
        c = c / (in_scale * t)
        out[i::2, k::2] = c
    if correct_colors is not None:
        M = np.asarray(correct_colors, dtype=np.float64).reshape(3, 3).astype(f32)
        o = np.empty_like(out)
        for r in range(3):  # mat3 @ vec3, sequential f32 dot
            o[..., r] = (M[r, 0] * out[..., 0] + M[r, 1] * out[..., 1]) + M[r, 2] * out[..., 2]
        out = o
    out = np.minimum(np.maximum(out, f32(0)), f32(1))
    return cast_out(out * out_scale, out_dtype)


def isp_color_matrix(correct_colors: bool, white_balance, color_correction):
    """camera_isp.py:360-369: cc.copy(); cc[:, :3] *= wb  (column j scaled by wb[j])."""
    if not correct_colors:
        return None
    cc = np.array(color_correction, dtype=np.float64).copy()
    cc[:, :3] *= np.asarray(white_balance, dtype=np.float64)
    return cc


DEFAULT_CC = np.array([[1.75, -0.25, -0.30], [-0.10, 1.40, -0.30], [-0.05, -0.55, 2.10]])  # :230-234
DEFAULT_WB = np.array([1.8, 1.0, 2.1])  # :245


# --------------------------------------------------------------------------------------
# interpolate.py
# --------------------------------------------------------------------------------------

TRANSFORMS = ["none", "rotate_90", "rotate_180", "rotate_270", "transpose", "flip_horiz",
              "flip_vert", "transverse"]  # interpolate.py:9-17


def resize_bilinear(src: np.ndarray, size, scale=None, dtype: str | None = None):
    """interpolate.py:19-34, 59-66, 128-139.  size = (w, h); scale scalar or (s0, s1) applied
    to (row, col).  scale=None reproduces the reference's crossed-axes quirk (:132-133)."""
    in_dtype = dtype_name(src)
    out_dtype = in_dtype if dtype is None else dtype
    Hs, Ws = src.shape[:2]
    if scale is None:
        scale = (size[0] / Hs, size[1] / Ws)
    if np.isscalar(scale):
        scale = (scale, scale)
    s0, s1 = f32(scale[0]), f32(scale[1])
    Wd, Hd = int(size[0]), int(size[1])
    intensity = f32(SCALE[out_dtype] / SCALE[in_dtype])

    pr = np.arange(Hd, dtype=np.int32).astype(f32) / s0
    pc = np.arange(Wd, dtype=np.int32).astype(f32) / s1
    ir, ic = np.trunc(pr).astype(np.int32), np.trunc(pc).astype(np.int32)
    fr, fc = (pr - ir.astype(f32))[:, None, None], (pc - ic.astype(f32))[None, :, None]
    r0, r1 = np.clip(ir, 0, Hs - 1), np.clip(ir + 1, 0, Hs - 1)
    c0, c1 = np.clip(ic, 0, Ws - 1), np.clip(ic + 1, 0, Ws - 1)
    s = src.astype(f32)

    def mix(x, y, a):  # taichi.math.mix
        return x * (f32(1.0) - a) + y * a

    y1 = mix(s[r0][:, c0], s[r1][:, c0], fr)
    y2 = mix(s[r0][:, c1], s[r1][:, c1], fr)
    out = mix(y1, y2, fc)
    return cast_out(out * intensity, out_dtype)


def py_round(x: float) -> int:
    return int(round(x))  # Python banker's rounding, as camera_isp.py:307,311


def isp_output_size(h: int, w: int, resize_width: int = 0, scale=None):
    """camera_isp.py:302-315 -> ((w_out, h_out), scale) or None when no resize."""
    if resize_width > 0:
        s = resize_width / w
        return (resize_width, py_round(h * s)), s
    if scale is not None:
        return (py_round(w * scale), py_round(h * scale)), scale
    return None


def transform(src: np.ndarray, name: str) -> np.ndarray:
    """interpolate.py:36-54, 93-125: dst[r,c] = src[transformed((Hd,Wd),(r,c))].

    'transverse' is not in the reference's dimension-swap list (:112-117) although its index
    map (:52) needs swapped dims, so on non-square images the reference reads out of bounds;
    it is only defined (here and in the product) for square images."""
    Hs, Ws = src.shape[:2]
    if name == "transverse":
        assert Hs == Ws, "transverse is only defined for square images (reference reads OOB otherwise)"
    swap = name in ("rotate_90", "rotate_270", "transpose")
    Hd, Wd = (Ws, Hs) if swap else (Hs, Ws)
    r, c = np.meshgrid(np.arange(Hd), np.arange(Wd), indexing="ij")
    if name == "rotate_90":
        sr, sc = Wd - c - 1, r
    elif name == "rotate_180":
        sr, sc = Hd - r - 1, Wd - c - 1
    elif name == "rotate_270":
        sr, sc = c, Hd - r - 1
    elif name == "transpose":
        sr, sc = c, r
    elif name == "flip_vert":
        sr, sc = Hd - r - 1, c
    elif name == "flip_horiz":
        sr, sc = r, Wd - c - 1
    elif name == "transverse":
        sr, sc = Wd - c - 1, Hd - r - 1
    else:
        sr, sc = r, c
    return np.ascontiguousarray(src[sr, sc])


# --------------------------------------------------------------------------------------
# camera_isp.py metering / tonemap (stateful ISP semantics)
# --------------------------------------------------------------------------------------

def _lerp(t, a, b):  # util.py:83-84
    return a + t * (b - a)


def metering_partials_bounds(images, stride: int = 8):
    """Phase 1 of camera_isp.py:142-154 on one shard: (min, max) of the subsample."""
    lo, hi = np.inf, -np.inf
    for im in images:
        s = im[::stride, ::stride, :].astype(f32)
        lo, hi = min(lo, float(s.min())), max(hi, float(s.max()))
    return np.array([lo, hi], dtype=f32)


def metering_partials_sums(images, b, stride: int = 8):
    """Phase 2 of camera_isp.py:117-128,159-162 on one shard given blended bounds b:
    returns [log_min, log_max, sum_log, sum_gray, sum_r, sum_g, sum_b] (f64) and n."""
    bmin, bmax = f32(b[0]), f32(b[1])
    lmin, lmax = np.inf, -np.inf
    sums = np.zeros(5, dtype=np.float64)
    n = 0
    for im in images:
        s = im[::stride, ::stride, :].astype(f32)
        sc = (s - bmin) / (bmax - bmin + f32(1e-6))
        g = (sc[..., 0] * GRAY_W[0] + sc[..., 1] * GRAY_W[1]) + sc[..., 2] * GRAY_W[2]
        lg = np.log(np.maximum(g, f32(1e-4)))
        lmin, lmax = min(lmin, float(np.nanmin(lg))), max(lmax, float(np.nanmax(lg)))
        sums += [lg.sum(dtype=np.float64), g.sum(dtype=np.float64),
                 sc[..., 0].sum(dtype=np.float64), sc[..., 1].sum(dtype=np.float64),
                 sc[..., 2].sum(dtype=np.float64)]
        n += g.size
    return np.array([lmin, lmax, *sums], dtype=np.float64), n


def metering_finish(prev, b, part, n, alpha):
    """camera_isp.py:131-134,164-166: normalise, then lerp the 9-vector with the previous
    state (the bounds are therefore blended twice -- reproduced on purpose)."""
    prev = np.asarray(prev, dtype=f32)
    alpha = f32(alpha)
    nn = f32(n)
    v = np.array([b[0], b[1], f32(part[0]), f32(part[1]), f32(part[2]) / nn, f32(part[3]) / nn,
                  f32(part[4]) / nn, f32(part[5]) / nn, f32(part[6]) / nn], dtype=f32)
    return (v + alpha * (prev - v)).astype(f32)


def metering_images(images, alpha, prev, stride: int = 8):
    """camera_isp.py:142-175.  images: list of (H,W,3); prev: f32[9]; returns new f32[9]."""
    prev = np.asarray(prev, dtype=f32)
    raw = metering_partials_bounds(images, stride)
    b = _lerp(f32(alpha), raw, prev[:2]).astype(f32)            # :156-157
    part, n = metering_partials_sums(images, b, stride)
    return metering_finish(prev, b, part, n, alpha)


class IspState:
    """The rolling state of camera_isp.ISP (camera_isp.py:267,376-385)."""

    def __init__(self, moving_alpha=0.1, stride=8):
        self.metrics = None
        self.moving_alpha = moving_alpha
        self.stride = stride

    def update_metering(self, images):
        if self.metrics is None:
            self.metrics = metering_images(images, 0.0, np.zeros(9, f32), self.stride)
        else:
            self.metrics = metering_images(images, 1.0 - self.moving_alpha, self.metrics, self.stride)
        return self.metrics


def reinhard_params(m, intensity, light_adapt, color_adapt):
    """Scalars of camera_isp.py:186-195."""
    m = np.asarray(m, dtype=f32)
    bmin, bmax, lmin, lmax, lmean, mean = m[:6]
    rgb_mean = m[6:9]
    with np.errstate(all="ignore"):
        key = (lmax - lmean) / (lmax - lmin)
        map_key = f32(0.3) + f32(0.7) * np.power(key, f32(1.4), dtype=f32)
    mean3 = (mean + f32(color_adapt) * (rgb_mean - mean)).astype(f32)
    return bmin, bmax, f32(map_key), mean3


def reinhard_isp(image: np.ndarray, m, gamma=1.0, intensity=1.0, light_adapt=1.0, color_adapt=0.0):
    """camera_isp.py:177-218.  Returns (u8 output, image-after-in-place-write-back)."""
    dt = dtype_name(image)
    bmin, bmax, map_key, mean3 = reinhard_params(m, intensity, light_adapt, color_adapt)
    ca, la = f32(color_adapt), f32(light_adapt)
    ei = f32(np.exp(f32(-intensity)))
    with np.errstate(all="ignore"):
        sc = (image.astype(f32) - bmin) / (bmax - bmin)
        g = ((sc[..., 0] * GRAY_W[0] + sc[..., 1] * GRAY_W[1]) + sc[..., 2] * GRAY_W[2])[..., None]
        ac = g + ca * (sc - g)
        am = mean3 + la * (ac - mean3)
        ad = np.power(ei * am, map_key, dtype=f32)
        p = sc * (f32(1.0) / (ad + sc))
        image_after = cast_out(p, dt)
        max_out = f32(max(1e-6, float(np.nanmax(p)) if np.any(~np.isnan(p)) else 1e-6))
        q = np.power(image_after.astype(f32) / max_out, f32(1.0 / gamma), dtype=f32)
        out = cast_out(f32(255) * q, "u8")
    return out, image_after


def linear_isp(image: np.ndarray, m, gamma=1.0):
    """camera_isp.py:220-227 -> tonemap.py:12-17 with the metering bounds, u8 out."""
    return _linear(image, f32(m[0]), f32(m[1]), gamma, "u8")


def _linear(image, lo, hi, gamma, dtype):
    with np.errstate(all="ignore"):
        inv = f32(1.0) / (hi - lo)
        x = np.power((image.astype(f32) - lo) * inv, f32(1.0) / f32(gamma), dtype=f32)
        x = np.fmin(np.fmax(x, f32(0)), f32(1))
        x = np.where(np.isnan(x), f32(0), x)
    return cast_out(x * f32(SCALE[dtype]), dtype)


# --------------------------------------------------------------------------------------
# tonemap.py (stateless)
# --------------------------------------------------------------------------------------

def bounds(image):
    """util.py:50-60."""
    x = image.astype(f32)
    return f32(np.nanmin(x)), f32(np.nanmax(x))


def tonemap_linear(src, gamma=1.0, dtype="u8"):
    """tonemap.py:27-46."""
    lo, hi = bounds(src)
    return _linear(src, lo, hi, gamma, dtype)


def stateless_metering(temp):
    """tonemap.py:78-103 on an image already in [0,1] (bounds 0..1).  Returns
    (B_min, B_max, log_mean, gray_mean, rgb_mean) with the reference's sign quirk
    B = (log_min, -log_max) (:102)."""
    sc = (temp - f32(0)) / (f32(1) - f32(0))
    g = (sc[..., 0] * GRAY_W[0] + sc[..., 1] * GRAY_W[1]) + sc[..., 2] * GRAY_W[2]
    lg = np.log(np.maximum(g, f32(1e-4)))
    n = f32(temp.shape[0] * temp.shape[1])
    lmin, lmax = f32(lg.min()), f32(lg.max())
    lmean = f32(lg.sum(dtype=np.float64)) / n
    gmean = f32(g.sum(dtype=np.float64)) / n
    rgb = np.array([f32(sc[..., k].sum(dtype=np.float64)) / n for k in range(3)], dtype=f32)
    return lmin, f32(-lmax), lmean, gmean, rgb


def tonemap_reinhard(src, gamma=1.0, intensity=1.0, light_adapt=1.0, color_adapt=0.0, dtype="u8",
                     return_intermediates=False):
    """tonemap.py:135-168: bounds -> normalise into f32 temp -> metering -> Reinhard in place
    on temp -> bounds of temp -> gamma/scale/cast."""
    ca, la = f32(color_adapt), f32(light_adapt)
    with np.errstate(all="ignore"):
        lo, hi = bounds(src)
        inv = f32(1.0) / (hi - lo)
        temp = np.fmin(np.fmax((src.astype(f32) - lo) * inv, f32(0)), f32(1))       # linear_func, gamma 1
        Bmin, Bmax, lmean, gmean, rgb_mean = stateless_metering(temp)
        key = (Bmax - lmean) / (Bmax - Bmin)
        map_key = f32(0.3) + f32(0.7) * np.power(key, f32(1.4), dtype=f32)
        mean3 = (gmean + ca * (rgb_mean - gmean)).astype(f32)
        ei = f32(np.exp(f32(-intensity)))
        g = ((temp[..., 0] * GRAY_W[0] + temp[..., 1] * GRAY_W[1]) + temp[..., 2] * GRAY_W[2])[..., None]
        ac = g + ca * (temp - g)
        am = mean3 + la * (ac - mean3)
        ad = np.power(ei * am, f32(map_key), dtype=f32)
        t2 = temp * (f32(1.0) / (ad + temp))
        lo2, hi2 = bounds(t2)
        out = _linear(t2, lo2, hi2, gamma, dtype)
    if return_intermediates:
        return out, dict(lo=lo, hi=hi, Bmin=Bmin, Bmax=Bmax, lmean=lmean, gmean=gmean,
                         rgb_mean=rgb_mean, map_key=f32(map_key), lo2=lo2, hi2=hi2)
    return out


# --------------------------------------------------------------------------------------
# whole-path compositions used by the tests and by bench.py's cpu_baseline leg
# --------------------------------------------------------------------------------------

def pipeline12_reinhard(packed, pattern=RGGB, ids_format=False, correct_colors=None, work="f16",
                        out="f16", gamma=1.0, intensity=1.0, light_adapt=1.0, color_adapt=0.0):
    """test/pipeline.py:26-32 (the stateless chain, BASELINE config 2):
    decode12(scaled, work dtype) -> bayer_to_rgb -> tonemap_reinhard(dtype=out)."""
    cfa = decode12(packed, work, scaled=True, ids_format=ids_format)
    rgb = bayer_to_rgb(cfa, pattern, correct_colors)
    return tonemap_reinhard(rgb, gamma, intensity, light_adapt, color_adapt, out)


def isp_load_packed12(packed, work="f16", pattern=RGGB, ids_format=False, correct_colors=None,
                      resize_width=0, scale=None):
    """camera_isp.py:333-340,371-373,302-315."""
    cfa = decode12(packed, work, scaled=True, ids_format=ids_format)
    rgb = bayer_to_rgb(cfa, pattern, correct_colors)
    sz = isp_output_size(rgb.shape[0], rgb.shape[1], resize_width, scale)
    return rgb if sz is None else resize_bilinear(rgb, sz[0], sz[1])


def isp_load_packed16(packed, work="f16", pattern=RGGB, correct_colors=None, resize_width=0, scale=None):
    """camera_isp.py:342-347."""
    cfa = decode16(packed, work, scaled=True)
    rgb = bayer_to_rgb(cfa, pattern, correct_colors)
    sz = isp_output_size(rgb.shape[0], rgb.shape[1], resize_width, scale)
    return rgb if sz is None else resize_bilinear(rgb, sz[0], sz[1])


# --------------------------------------------------------------------------------------
# color/yuv_420.py  (the step after the path)
# --------------------------------------------------------------------------------------
# yuv_420.py:12-16: the matrix is named for BGR input but rgb_YCrCb feeds it rgb.bgr (:26-27), so
# Y = 0.299 B + 0.587 G + 0.114 R for an RGB image -- reproduced as written.
YCRCB_T_BGR = np.array([[0.299, 0.587, 0.114], [-0.168736, -0.331264, 0.5], [0.5, -0.418688, -0.081312]])
BGR_T_YCRCB = np.linalg.inv(YCRCB_T_BGR)          # yuv_420.py:18 (Python-scope inverse, float64)


def _matvec3(m, v0, v1, v2):
    """Taichi mat @ vec in f32: (m0*v0 + m1*v1) + m2*v2 per row."""
    m = m.astype(f32)
    return [(m[i, 0] * v0 + m[i, 1] * v1) + m[i, 2] * v2 for i in range(3)]


def _clamp_quirk(x):
    """tm.clamp(0, 1, x) (yuv_420.py:54,57,88): the arguments are (x=0, xmin=1, xmax=x), i.e.
    min(max(0, 1), x) = min(1, x) -- no lower clamp."""
    return np.minimum(f32(1.0), x)


def split_yuv_420(yuv):
    """yuv_420.py:95-103."""
    height = yuv.shape[0] * 2 // 3
    width = yuv.shape[1]
    return yuv[:height], yuv[height:].reshape(2, height // 2, width // 2), (width, height)


def rgb_yuv420(src: np.ndarray, dtype: str | None = None) -> np.ndarray:
    """rgb_yuv420_kernel + rgb_yuv420_image (yuv_420.py:39-66,105-119)."""
    in_dt = dtype_name(src)
    out_dt = in_dt if dtype is None else dtype
    H, W, _ = src.shape
    x = src.astype(f32) / f32(SCALE[in_dt])                    # :52  src / in_scale
    y0, u0, v0 = _matvec3(YCRCB_T_BGR, x[..., 2], x[..., 1], x[..., 0])   # rgb.bgr
    u0 = u0 + f32(0.5); v0 = v0 + f32(0.5)                      # :23  + vec3(0, 0.5, 0.5) (y + 0 is exact)
    yuv = np.zeros((H * 3 // 2, W), dtype=NP_DTYPE[out_dt])
    yp, uvp, _ = split_yuv_420(yuv)
    He, We = H // 2 * 2, W // 2 * 2                             # the loop covers whole 2x2 blocks (:48)
    yp[:He, :We] = cast_out(_clamp_quirk(y0[:He, :We]) * f32(SCALE[out_dt]), out_dt)     # :54
    acc_u = np.zeros((He // 2, We // 2), dtype=f32); acc_v = np.zeros_like(acc_u)
    for dr, dc in ((0, 0), (0, 1), (1, 0), (1, 1)):            # ti.ndrange(2, 2) order, sequential f32 adds (:55)
        acc_u = acc_u + u0[dr:He:2, dc:We:2]
        acc_v = acc_v + v0[dr:He:2, dc:We:2]
    uvp[1] = cast_out(_clamp_quirk(acc_u / f32(4.0)) * f32(SCALE[out_dt]), out_dt)         # :57-58
    uvp[0] = cast_out(_clamp_quirk(acc_v / f32(4.0)) * f32(SCALE[out_dt]), out_dt)         # :59
    return yuv


def yuv420_rgb(yuv: np.ndarray, dtype: str | None = None) -> np.ndarray:
    """yuv420_rgb_kernel + yuv420_rgb_image (yuv_420.py:68-92,121-131)."""
    in_dt = dtype_name(yuv)
    out_dt = in_dt if dtype is None else dtype
    yp, uvp, (W, H) = split_yuv_420(yuv)
    inv = f32(SCALE[in_dt])
    rr, cc = np.arange(H)[:, None] // 2, np.arange(W)[None, :] // 2
    y = yp.astype(f32) / inv
    u = uvp[1][rr, cc].astype(f32) / inv - f32(0.5)             # :84-88  (yuv / in_scale) - (0, .5, .5)
    v = uvp[0][rr, cc].astype(f32) / inv - f32(0.5)
    b, g, r = _matvec3(BGR_T_YCRCB, y, u, v)                    # YCrCb_bgr, then .bgr (:30-35)
    rgb = np.stack([r, g, b], axis=-1)
    return cast_out(_clamp_quirk(rgb) * f32(SCALE[out_dt]), out_dt)
