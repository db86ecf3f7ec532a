"""Stateful camera ISP -- call surface of taichi_image/camera_isp.py (Camera16 / Camera32).

load (unpack/normalise) -> demosaic(+colour matrix) -> resize; rolling metering statistics;
Reinhard / linear tonemap to u8; orientation transform.  All device work is HIP
(csrc/): the load path is one fused tile kernel over the packed frame, the tonemaps are
two-pass elementwise kernels with wave-shuffle reductions.

Deliberate differences from the reference (see DESIGN.md "quirks"):
 * `_process_image` forwards `self.bayer_pattern` (the reference drops it and always demosaics
   RGGB, camera_isp.py:372); identical for RGGB.  `reference_quirks=True` reproduces the reference.
 * tonemap parameters are runtime floats (the reference re-JITs per value).
 * NaN / out-of-range float->u8 casts are defined (0 / saturate) where the reference is undefined.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch

from . import _native, bayer, interpolate, packed, types
from . import distributed as _dist

default_cc = np.array([      # camera_isp.py:230-234
    [1.75, -0.25, -0.30],
    [-0.10, 1.40, -0.30],
    [-0.05, -0.55, 2.10],
])


def _typecheck(name, value, kinds, optional=False):
    if optional and value is None:
        return
    if isinstance(value, bool) and bool not in (kinds if isinstance(kinds, tuple) else (kinds,)):
        raise TypeError(f"{name} must be {kinds}, got bool")
    if not isinstance(value, kinds):
        raise TypeError(f"{name} must be {kinds}, got {type(value).__name__}")


class MeteringTimeout(RuntimeError):
    """The grid barrier of an earlier one-launch update_metering on this device timed out (something else held compute
    units of the GPU for about a second).  That call left the metrics as they were - bounds folded from half of the blocks
    must never enter the rolling average - so everything tone-mapped with them since is suspect.  Raised by the NEXT
    update_metering / tonemap call (one host read of the device's mailbox, no synchronisation)."""


class TonemapTimeout(RuntimeError):
    """The grid-wide wait of an earlier one-launch tonemap_reinhard on this device (max_out, camera_isp.py:213) timed out:
    the u8 outputs of that call are invalid.  Raised by the next metering / tonemap call (a host read of the mailbox)."""


def _version_of(t):
    """torch's write counter of a tensor, None where there is none (inference tensors do not track one)."""
    try:
        return t._version
    except RuntimeError:
        return None


def _tag_subsample(rgb, sub, stride):
    """Hang the dense metering subsample the load kernel left (`rgb[::stride, ::stride]`) on the image.  The tag is valid
    while the image's version counter stands still: torch writes move it, the library's own in-place writes move it
    explicitly (_written_in_place).  A tensor without a counter (created under torch.inference_mode(), as the reference's
    bench does, bench/camera_isp.py:53) gets no tag - update_metering then gathers from the image itself."""
    v = _version_of(rgb)
    if v is not None:
        rgb._mi_metering_sub = (sub, stride, v)


def _valid_subsample(im, stride):
    tag = getattr(im, "_mi_metering_sub", None)
    if tag is None or tag[1] != stride or tag[2] != _version_of(im):
        return None
    return tag[0]


def _written_in_place(images):
    """The library has overwritten these images through their raw pointers (Reinhard pass 1 writes the mapped values p
    back, camera_isp.py:211): what torch cannot see is made visible - the version counter moves (views and autograd see
    a write) and a metering subsample taken before the write is dropped, so the next update_metering meters the
    MUTATED image as the reference does (camera_isp.py:168-175 on the image of :211)."""
    for im in images:
        if getattr(im, "_mi_metering_sub", None) is not None:
            del im._mi_metering_sub
        if _version_of(im) is not None:
            torch.autograd.graph.increment_version(im)


def camera_isp(name: str, dtype=types.f32):
    """camera_isp.py:75-418: class factory specialised on the working dtype."""
    dtype = types.as_dtype(dtype)
    torch_dtype = types.ti_to_torch[dtype]

    def _check_image(image, what="image"):
        if not isinstance(image, torch.Tensor):
            raise TypeError(f"{what} must be a torch.Tensor")
        assert image.ndim == 3 and image.shape[2] == 3, f"{what} must be (H, W, 3)"
        assert image.dtype == torch_dtype, f"{what} must be {torch_dtype}, got {image.dtype}"
        assert image.is_cuda and image.is_contiguous(), f"{what} must be a contiguous device tensor"

    def reinhard_kernel(image, output, metering, gamma, intensity, light_adapt, color_adapt,
                        transform=interpolate.ImageTransform.none):
        """camera_isp.py:177-218.  Mutates `image` (p written back in place), fills `output`."""
        _check_image(image)
        H, W = image.shape[:2]
        ws = _native.workspace(H, W, image.device)
        _native.check(_native.lib().mi_isp_reinhard(
            image.data_ptr(), output.data_ptr(), H, W, dtype.code, metering.data_ptr(), float(gamma),
            float(intensity), float(light_adapt), float(color_adapt), interpolate.transform_code(transform),
            ws.data_ptr(), _native.stream_ptr(image.device)))
        _written_in_place([image])

    def linear_kernel(image, output, metering, gamma, transform=interpolate.ImageTransform.none):
        """camera_isp.py:220-227."""
        _check_image(image)
        H, W = image.shape[:2]
        ws = _native.workspace(H, W, image.device)
        _native.check(_native.lib().mi_isp_linear(
            image.data_ptr(), output.data_ptr(), H, W, dtype.code, metering.data_ptr(), float(gamma),
            interpolate.transform_code(transform), ws.data_ptr(), _native.stream_ptr(image.device)))

    def _out_shape(image, transform):
        H, W = image.shape[:2]
        if transform in (interpolate.ImageTransform.rotate_90, interpolate.ImageTransform.rotate_270,
                         interpolate.ImageTransform.transpose):
            return (W, H, 3)
        return (H, W, 3)

    class ISP():
        def __init__(self, bayer_pattern: bayer.BayerPattern,
                     scale: Optional[float] = None,
                     resize_width: int = 0,
                     moving_alpha=0.1,
                     correct_colors: bool = False,
                     white_balance: np.ndarray = np.array([1.8, 1.0, 2.1]),
                     color_correction: np.ndarray = default_cc,
                     transform: interpolate.ImageTransform = interpolate.ImageTransform.none,
                     device: torch.device = torch.device('cuda', 0),
                     metering_stride: int = 8,
                     process_group=None,
                     reference_quirks: bool = False):
            _typecheck("bayer_pattern", bayer_pattern, bayer.BayerPattern)
            _typecheck("scale", scale, float, optional=True)
            _typecheck("resize_width", resize_width, int)
            _typecheck("correct_colors", correct_colors, bool)
            _typecheck("white_balance", white_balance, np.ndarray)
            _typecheck("color_correction", color_correction, np.ndarray)
            _typecheck("transform", transform, interpolate.ImageTransform)
            _typecheck("device", device, torch.device)
            _typecheck("metering_stride", metering_stride, int)
            assert scale is None or resize_width == 0, "Cannot specify both scale and resize_width"

            self.bayer_pattern = bayer_pattern
            # reference_quirks=True: demosaic as the reference does - ISP._process_image calls bayer_to_rgb WITHOUT its
            # pattern (camera_isp.py:371-373), so every camera is demosaiced as RGGB whatever bayer_pattern says.  For
            # comparisons against the reference's own outputs; the default honours the pattern.
            _typecheck("reference_quirks", reference_quirks, bool)
            self.reference_quirks = reference_quirks
            self.moving_alpha = moving_alpha
            self.scale = scale
            self.resize_width = resize_width
            self.transform = transform
            self.metering_stride = metering_stride

            self.correct_colors = correct_colors
            self.white_balance = white_balance
            self.color_correction = color_correction

            self.metrics = None
            self.device = device
            # one-process-per-GPU sharding: statistics are all-reduced over this group (RCCL)
            self.process_group = process_group

        @property
        def _demosaic_pattern(self):
            return bayer.BayerPattern.RGGB if self.reference_quirks else self.bayer_pattern

        def set(self, moving_alpha: Optional[float] = None, resize_width: Optional[int] = None,
                scale: Optional[float] = None,
                correct_colors: Optional[bool] = None,
                white_balance: Optional[np.ndarray] = None,
                color_correction: Optional[np.ndarray] = None,
                transform: Optional[interpolate.ImageTransform] = None):
            """camera_isp.py:270-300."""
            _typecheck("moving_alpha", moving_alpha, float, optional=True)
            _typecheck("resize_width", resize_width, int, optional=True)
            _typecheck("scale", scale, float, optional=True)
            _typecheck("correct_colors", correct_colors, bool, optional=True)
            _typecheck("white_balance", white_balance, np.ndarray, optional=True)
            _typecheck("color_correction", color_correction, np.ndarray, optional=True)
            _typecheck("transform", transform, interpolate.ImageTransform, optional=True)
            if moving_alpha is not None:
                self.moving_alpha = moving_alpha
            if resize_width is not None:
                self.resize_width = resize_width
                self.scale = None
            if scale is not None:
                self.scale = scale
                self.resize_width = 0
            if transform is not None:
                self.transform = transform
            if correct_colors is not None:
                self.correct_colors = correct_colors
            if white_balance is not None:
                self.white_balance = white_balance
            if color_correction is not None:
                self.color_correction = color_correction

        def resize_image(self, image):
            """camera_isp.py:302-315."""
            w, h = image.shape[1], image.shape[0]
            if self.resize_width > 0:
                scale = self.resize_width / w
                output_size = (self.resize_width, round(h * scale))
                return interpolate.resize_bilinear(image, output_size, scale)
            elif self.scale is not None:
                output_size = (round(w * self.scale), round(h * self.scale))
                return interpolate.resize_bilinear(image, output_size, self.scale)
            else:
                return image

        def _convert(self, image, mode, src_dtype):
            if not isinstance(image, torch.Tensor):
                raise TypeError("image must be a torch.Tensor")
            assert image.ndim == 2, "image must be a 2-D CFA"
            assert image.dtype == src_dtype, f"image must be {src_dtype}, got {image.dtype}"
            src = image.to(self.device).contiguous()
            cfa = torch.empty(image.shape, dtype=torch_dtype, device=self.device)
            _native.check(_native.lib().mi_isp_load_convert(src.data_ptr(), cfa.data_ptr(), cfa.numel(), mode,
                                                            dtype.code, _native.stream_ptr(self.device)))
            return self._process_image(cfa)

        def load_16u(self, image):
            """camera_isp.py:318-321 (kernel :82-87)."""
            return self._convert(image, 0, torch.uint16)

        def load_16f(self, image):
            """camera_isp.py:323-326 (kernel :95-99: u16 converted numerically)."""
            return self._convert(image, 2, torch.uint16)

        def load_32f(self, image):
            """camera_isp.py:328-331 (kernel :89-93)."""
            return self._convert(image, 1, torch.float32)

        def _load_packed(self, image_data, bits, ids_format):
            if not isinstance(image_data, torch.Tensor):
                raise TypeError("image_data must be a torch.Tensor")
            assert image_data.ndim == 2 and image_data.dtype == torch.uint8, "image_data must be (H, bytes) uint8"
            if bits == 12:
                assert image_data.shape[1] % 3 == 0, "packed-12 rows must hold whole pixel pairs (bytes % 3 == 0)"
                w, h = (image_data.shape[1] * 2 // 3, image_data.shape[0])        # camera_isp.py:336
            else:
                w, h = (image_data.shape[1] // 2, image_data.shape[0])             # camera_isp.py:343
            assert w % 2 == 0 and h % 2 == 0, "image must be even size"
            src = image_data.to(self.device).contiguous()
            L = _native.lib()
            # camera_isp.py:302-312: output size and scale of resize_image
            if self.resize_width > 0:
                scale = self.resize_width / w
                out_size = (self.resize_width, round(h * scale))
            elif self.scale is not None:
                scale = self.scale
                out_size = (round(w * scale), round(h * scale))
            else:
                scale, out_size = 0.0, (w, h)
            fused = scale > 0 and min(out_size) > 0 and L.mi_isp_load_packed_scale_supported(float(scale))
            wd, hd = out_size if fused else (w, h)
            rgb = torch.empty((hd, wd, 3), dtype=torch_dtype, device=self.device)
            if not fused and scale > 0:                  # a scale the fused kernel does not take: resize separately
                _native.check(L.mi_isp_load_packed(
                    src.data_ptr(), rgb.data_ptr(), h, w, bits, int(bool(ids_format)), self._demosaic_pattern.value,
                    _native.ccm_arg(self.color_correct_matrix), dtype.code, hd, wd, 0.0, _native.stream_ptr(self.device)))
                return self.resize_image(rgb)
            st = self.metering_stride
            if fused or not L.mi_isp_load_packed_metered_is_fused(h, w, bits, int(bool(ids_format)), dtype.code, st):
                _native.check(L.mi_isp_load_packed(
                    src.data_ptr(), rgb.data_ptr(), h, w, bits, int(bool(ids_format)), self._demosaic_pattern.value,
                    _native.ccm_arg(self.color_correct_matrix), dtype.code, hd, wd, float(scale) if fused else 0.0,
                    _native.stream_ptr(self.device)))
                return rgb
            # the image and, on the way, the stride-subsampled copy update_metering will ask for (camera_isp.py:168-170):
            # the load kernel holds those pixels anyway, the strided gather over six 4K images costs 25 us per call
            sub = torch.empty(((hd + st - 1) // st, (wd + st - 1) // st, 3), dtype=torch_dtype, device=self.device)
            _native.check(L.mi_isp_load_packed_metered(
                src.data_ptr(), rgb.data_ptr(), h, w, bits, int(bool(ids_format)), self._demosaic_pattern.value,
                _native.ccm_arg(self.color_correct_matrix), dtype.code, hd, wd, 0.0,
                sub.data_ptr(), st, _native.stream_ptr(self.device)))
            _tag_subsample(rgb, sub, st)
            return rgb

        def load_packed12(self, image_data, ids_format=False):
            """camera_isp.py:333-340: unpack + demosaic (+ccm) fused in one pass over the packed frame."""
            return self._load_packed(image_data, 12, ids_format)

        def load_packed12_batch(self, images_data: List[torch.Tensor], ids_format=False) -> List[torch.Tensor]:
            """Extension (not in the reference): `[self.load_packed12(d, ids_format) for d in images_data]` for the cameras
            of one group - frames of one size - in ONE launch per 8 cameras (mi_isp_load_packed_batch): same results, bit
            for bit, without the other launches' dispatch, table build and drain (config 3: 43.0 -> 39.5 us per frame)."""
            return self._load_packed_batch(images_data, 12, ids_format)

        def load_packed16_batch(self, images_data: List[torch.Tensor]) -> List[torch.Tensor]:
            """The same for `load_packed16` (camera_isp.py:342-347)."""
            return self._load_packed_batch(images_data, 16, False)

        def _load_packed_batch(self, images_data, bits, ids_format):
            _typecheck("images_data", images_data, list)
            if len(images_data) == 0:
                return []
            for d in images_data:
                if not isinstance(d, torch.Tensor):
                    raise TypeError("image_data must be a torch.Tensor")
                assert d.ndim == 2 and d.dtype == torch.uint8, "image_data must be (H, bytes) uint8"
                assert d.shape == images_data[0].shape, "the frames of a batch must share a shape"
            h = images_data[0].shape[0]
            w = images_data[0].shape[1] * 2 // 3 if bits == 12 else images_data[0].shape[1] // 2
            if bits == 12:
                assert images_data[0].shape[1] % 3 == 0, "packed-12 rows must hold whole pixel pairs (bytes % 3 == 0)"
            assert w % 2 == 0 and h % 2 == 0, "image must be even size"
            L = _native.lib()
            if self.resize_width > 0:
                scale = self.resize_width / w
                out_size = (self.resize_width, round(h * scale))
            elif self.scale is not None:
                scale = self.scale
                out_size = (round(w * scale), round(h * scale))
            else:
                scale, out_size = 0.0, (w, h)
            fused = scale > 0 and min(out_size) > 0 and L.mi_isp_load_packed_scale_supported(float(scale))
            if scale > 0 and not fused:                      # a scale the fused kernel does not take
                return [self._load_packed(d, bits, ids_format) for d in images_data]
            wd, hd = out_size if fused else (w, h)
            srcs = [d.to(self.device).contiguous() for d in images_data]
            rgbs = [torch.empty((hd, wd, 3), dtype=torch_dtype, device=self.device) for _ in srcs]
            st = self.metering_stride
            metered = not fused and bool(L.mi_isp_load_packed_metered_is_fused(h, w, bits, int(bool(ids_format)), dtype.code, st))
            subs = [torch.empty(((hd + st - 1) // st, (wd + st - 1) // st, 3), dtype=torch_dtype, device=self.device)
                    for _ in srcs] if metered else None
            _native.check(L.mi_isp_load_packed_batch(
                _native.ptr_array(srcs), _native.ptr_array(rgbs), None if subs is None else _native.ptr_array(subs), len(srcs),
                h, w, bits, int(bool(ids_format)), self._demosaic_pattern.value, _native.ccm_arg(self.color_correct_matrix),
                dtype.code, hd, wd, float(scale) if fused else 0.0, st, _native.stream_ptr(self.device)))
            if subs is not None:
                for rgb, sub in zip(rgbs, subs):
                    _tag_subsample(rgb, sub, st)
            return rgbs

        def load_packed16(self, image_data):
            """camera_isp.py:342-347."""
            return self._load_packed(image_data, 16, False)

        @property
        def color_correct_matrix(self) -> Optional[np.ndarray]:
            """camera_isp.py:360-369: cc with column j scaled by white_balance[j]."""
            if self.correct_colors:
                cc = self.color_correction.copy()
                cc[:, :3] *= self.white_balance
                return cc
            return None

        def _process_image(self, cfa):
            """camera_isp.py:371-373."""
            rgb = bayer.bayer_to_rgb(cfa, pattern=self._demosaic_pattern, correct_colors=self.color_correct_matrix)
            return self.resize_image(rgb)

        def _metering_images(self, images, t, prev, stride=None):
            """camera_isp.py:168-175: statistics of the stride-subsampled images, blended into a
            copy of `prev`; the subsample is gathered in-kernel (no torch.stack copy)."""
            assert len(images) > 0, "need at least one image"
            for im in images:
                _check_image(im)
                assert im.shape == images[0].shape, "all images of one call must share a shape"
            H, W = images[0].shape[:2]
            ws = _native.workspace(H, W, self.device)
            stride = self.metering_stride if stride is None else stride       # (stride=1: the caller hands over subsamples)
            # images that came out of load_packed12 / 16 carry their subsample: the same samples in the same order from a
            # dense buffer (stride 1) - identical results, no strided gather over the full-size images
            subs = [_valid_subsample(im, stride) for im in images]
            if all(s is not None for s in subs):
                images = subs
                H, W = images[0].shape[:2]
                stride = 1
            ptrs = _native.ptr_array(images)
            L = _native.lib()
            stream = _native.stream_ptr(self.device)
            with torch.cuda.device(self.device):
                if L.mi_isp_metering_faults(1):
                    raise MeteringTimeout("an earlier update_metering on this device timed out at its grid barrier: its "
                                          "metrics were left unchanged and outputs tone-mapped with them are invalid")
                if L.mi_isp_reinhard_faults(1):
                    raise TonemapTimeout("an earlier tonemap_reinhard on this device timed out waiting for an image's "
                                         "max_out: the outputs of that call are invalid")
                if L.mi_isp_camera_group_faults(1):
                    raise TonemapTimeout("an earlier process_packed12 on this device timed out waiting for an image's "
                                         "max_out: the outputs of that call are invalid")
            if self.process_group is None:
                # (the reference clones `prev` and lets the kernel update the clone, camera_isp.py:172-173; here the kernel
                # reads `prev` and writes the new tensor: no copy kernel - 4 us - in front of every update)
                metering = torch.empty_like(prev)
                _native.check(L.mi_isp_metering_to(ptrs, len(images), H, W, stride, dtype.code, prev.data_ptr(),
                                                   metering.data_ptr(), float(t), ws.data_ptr(), stream))
                return metering
            # sharded batch: the same two data passes, an all-gather after each (two collectives per call), the ranks'
            # rows combined by one small kernel each on this stream (mi_isp_metering_combine_*)
            world = _dist.world_size(self.process_group)
            raw = torch.empty(2, dtype=torch.float32, device=self.device)
            _native.check(L.mi_isp_metering_bounds(ptrs, len(images), H, W, stride, dtype.code,
                                                   raw.data_ptr(), ws.data_ptr(), stream))
            gathered = _dist.all_gather_rows(raw, self.process_group)
            b = torch.empty(2, dtype=torch.float32, device=self.device)
            _native.check(L.mi_isp_metering_combine_bounds(gathered.data_ptr(), world, prev.data_ptr(), float(t),
                                                           b.data_ptr(), stream))
            part = torch.empty(8, dtype=torch.float32, device=self.device)
            _native.check(L.mi_isp_metering_sums(ptrs, len(images), H, W, stride, dtype.code,
                                                 b.data_ptr(), part.data_ptr(), ws.data_ptr(), stream))
            gathered8 = _dist.all_gather_rows(part, self.process_group)
            metering = prev.clone()
            _native.check(L.mi_isp_metering_combine_sums(gathered8.data_ptr(), world, b.data_ptr(), metering.data_ptr(),
                                                         float(t), stream))
            return metering

        def update_metering(self, images: List[torch.Tensor]):
            """camera_isp.py:376-385."""
            if self.metrics is None:
                initial = torch.zeros(9, dtype=torch.float32, device=self.device)
                self.metrics = self._metering_images(images, 0.0, initial)
            else:
                self.metrics = self._metering_images(images, (1.0 - self.moving_alpha), self.metrics)

        def tonemap_only(self, image, metrics, gamma, intensity, light_adapt, color_adapt):
            """camera_isp.py:387-390."""
            output = torch.empty(_out_shape(image, self.transform), dtype=torch.uint8, device=self.device)
            reinhard_kernel(image, output, metrics, gamma, intensity, light_adapt, color_adapt, self.transform)
            return output

        def tonemap_reinhard(self, images: List[torch.Tensor],
                             gamma: float = 1.0, intensity: float = 1.0, light_adapt: float = 1.0,
                             color_adapt: float = 0.0, write_back: bool = True):
            """camera_isp.py:394-403.  NOTE: like the reference, pass 1 overwrites each input image
            with the Reinhard-mapped values (camera_isp.py:211).
            write_back=False (an extension, not the reference's semantics): the same u8 outputs, bit for bit, with the
            images left as they are - a third of the tonemap's memory traffic is that write and its re-read."""
            _typecheck("write_back", write_back, bool)
            _typecheck("images", images, list)
            for n, v in (("gamma", gamma), ("intensity", intensity), ("light_adapt", light_adapt),
                         ("color_adapt", color_adapt)):
                _typecheck(n, v, float)
            self.update_metering(images)
            outputs = [torch.empty(_out_shape(image, self.transform), dtype=torch.uint8, device=self.device)
                       for image in images]
            # one batched call for the whole list (the reference loops, camera_isp.py:400-401); the
            # orientation transform (:403) is folded into the u8 store
            H, W = images[0].shape[:2]
            ws = _native.workspace(H, W, self.device)
            fn = _native.lib().mi_isp_reinhard_batch if write_back else _native.lib().mi_isp_reinhard_batch_keep
            _native.check(fn(
                _native.ptr_array(images), _native.ptr_array(outputs), len(images), H, W, dtype.code,
                self.metrics.data_ptr(), float(gamma), float(intensity), float(light_adapt), float(color_adapt),
                interpolate.transform_code(self.transform), ws.data_ptr(), _native.stream_ptr(self.device)))
            if write_back:
                _written_in_place(images)
            return outputs

        def tonemap_reinhard_yuv420(self, images: List[torch.Tensor],
                                    gamma: float = 1.0, intensity: float = 1.0, light_adapt: float = 1.0,
                                    color_adapt: float = 0.0):
            """Extension (not in the reference): `[color.rgb_yuv420_image(o) for o in tonemap_reinhard(images, ...)]`
            - planar YUV 4:2:0 u8 `(H * 3 / 2, W)` per image for video encoders - with the conversion
            (color/yuv_420.py:39-66) fused into the second Reinhard pass when no orientation transform is set
            and W % 16 == 0: the u8 RGB images are never written.  Same side effects as tonemap_reinhard."""
            from . import color
            _typecheck("images", images, list)
            H, W = images[0].shape[:2]
            if self.transform != interpolate.ImageTransform.none or H % 2 or W % 16:
                return [color.rgb_yuv420_image(o) for o in self.tonemap_reinhard(images, gamma, intensity,
                                                                                  light_adapt, color_adapt)]
            for n, v in (("gamma", gamma), ("intensity", intensity), ("light_adapt", light_adapt),
                         ("color_adapt", color_adapt)):
                _typecheck(n, v, float)
            self.update_metering(images)
            outputs = [torch.empty((H * 3 // 2, W), dtype=torch.uint8, device=self.device) for _ in images]
            ws = _native.workspace(H, W, self.device)
            _native.check(_native.lib().mi_isp_reinhard_batch_yuv420(
                _native.ptr_array(images), _native.ptr_array(outputs), len(images), H, W, dtype.code,
                self.metrics.data_ptr(), float(gamma), float(intensity), float(light_adapt), float(color_adapt),
                ws.data_ptr(), _native.stream_ptr(self.device)))
            _written_in_place(images)
            return outputs

        def process_packed12(self, frames: List[torch.Tensor], gamma: float = 1.0, intensity: float = 1.0,
                             light_adapt: float = 1.0, color_adapt: float = 0.0, keep_images: bool = False,
                             ids_format: bool = False):
            """Extension (not in the reference): one step of the reference's own bench in one call -
            `Processor.__call__` of bench/camera_isp.py:23-27:

                images = [isp.load_packed12(f, ids_format) for f in frames]
                return isp.tonemap_reinhard(images, gamma=...)

            with the same u8 outputs and the same metering state afterwards, bit for bit.  For a full-resolution
            Camera16 group that fits the chip (`mi_isp_camera_group_fits`: 4096 x 3072 on MI355X, metering stride 8, no
            resize, no orientation transform, single process) the loaded images never exist in memory: the metering reads
            a subsample demosaiced straight from the packed frames, and ONE persistent launch takes every camera from
            packed bytes to its u8 image (csrc/isp_mega_cam.h).  Everything else takes the two calls above.
            keep_images=True returns `(outputs, images)`, the images holding what the reference leaves in them (p,
            camera_isp.py:211); by default only the outputs are returned, as the bench's Processor does."""
            _typecheck("frames", frames, list)
            _typecheck("keep_images", keep_images, bool)
            for n, v in (("gamma", gamma), ("intensity", intensity), ("light_adapt", light_adapt),
                         ("color_adapt", color_adapt)):
                _typecheck(n, v, float)
            assert len(frames) > 0, "need at least one frame"
            L = _native.lib()
            f0 = frames[0]
            fused = (dtype is types.f16 and not ids_format and self.resize_width == 0 and self.scale is None
                     and self.transform == interpolate.ImageTransform.none and self.metering_stride == 8
                     and 1 <= len(frames) <= 64
                     and all(isinstance(f, torch.Tensor) and f.ndim == 2 and f.dtype == torch.uint8 and f.shape == f0.shape
                             for f in frames)
                     and f0.shape[1] % 3 == 0)
            if fused and torch.cuda.is_current_stream_capturing():
                fused = False                  # (a captured resident launch can be neither ordered against others nor checked)
            if fused:
                h, w = f0.shape[0], f0.shape[1] * 2 // 3
                with torch.cuda.device(self.device):
                    fused = bool(L.mi_isp_camera_group_fits(h, w, self._demosaic_pattern.value, dtype.code, 8))
            if not fused:
                images = self.load_packed12_batch(frames, ids_format)
                outputs = self.tonemap_reinhard(images, gamma, intensity, light_adapt, color_adapt)
                return (outputs, images) if keep_images else outputs
            srcs = [f.to(self.device).contiguous() for f in frames]
            with torch.cuda.device(self.device):
                if L.mi_isp_metering_faults(1):
                    raise MeteringTimeout("an earlier update_metering on this device timed out at its grid barrier: its "
                                          "metrics were left unchanged and outputs tone-mapped with them are invalid")
                if L.mi_isp_camera_group_faults(1):
                    raise TonemapTimeout("an earlier process_packed12 on this device timed out waiting for an image's "
                                         "max_out: the outputs of that call are invalid")
            n = len(srcs)
            outputs = [torch.empty((h, w, 3), dtype=torch.uint8, device=self.device) for _ in srcs]
            images = [torch.empty((h, w, 3), dtype=torch_dtype, device=self.device) for _ in srcs] if keep_images else None
            if self.metrics is None:                         # camera_isp.py:376-385
                prev, t = torch.zeros(9, dtype=torch.float32, device=self.device), 0.0
            else:
                prev, t = self.metrics, 1.0 - self.moving_alpha
            metrics = torch.empty_like(prev)                 # (the previous state is read, the new one written: no clone)
            scratch = torch.empty(int(L.mi_isp_camera_group_scratch_bytes(n, h, w)), dtype=torch.uint8, device=self.device)
            ws = _native.workspace(h, w, self.device, slots=n + 1)
            stream = _native.stream_ptr(self.device)
            p_srcs, p_imgs, p_outs = _native.ptr_array(srcs), _native.ptr_array(images) if keep_images else None, _native.ptr_array(outputs)
            ccm = _native.ccm_arg(self.color_correct_matrix)
            if self.process_group is None:
                _native.check(L.mi_isp_camera_group_reinhard(
                    p_srcs, p_imgs, p_outs, n, h, w, self._demosaic_pattern.value, ccm, prev.data_ptr(), metrics.data_ptr(), float(t),
                    float(gamma), float(intensity), float(light_adapt), float(color_adapt), scratch.data_ptr(), ws.data_ptr(),
                    stream))
                self.metrics = metrics
                return (outputs, images) if keep_images else outputs
            # a sharded group (one process per GPU): the same three steps with the metering's two all-gathers in between
            _native.check(L.mi_isp_camera_group_subsample(p_srcs, n, h, w, self._demosaic_pattern.value, ccm,
                                                          scratch.data_ptr(), stream))
            per = int(L.mi_isp_camera_group_scratch_bytes(1, h, w))
            hs, ws_ = (h + 7) // 8, (w + 7) // 8
            subs = [scratch[i * per:i * per + hs * ws_ * 6].view(torch_dtype).view(hs, ws_, 3) for i in range(n)]
            self.metrics = self._metering_images(subs, t, prev, stride=1)
            _native.check(L.mi_isp_camera_group_tonemap(
                p_srcs, p_imgs, p_outs, n, h, w, self._demosaic_pattern.value, ccm, self.metrics.data_ptr(), float(gamma),
                float(intensity), float(light_adapt), float(color_adapt), ws.data_ptr(), stream))
            return (outputs, images) if keep_images else outputs

        def tonemap_linear(self, images: List[torch.Tensor], gamma: float = 1.0):
            """camera_isp.py:405-413."""
            _typecheck("images", images, list)
            _typecheck("gamma", gamma, float)
            self.update_metering(images)
            outputs = [torch.empty(_out_shape(image, self.transform), dtype=torch.uint8, device=self.device)
                       for image in images]
            H, W = images[0].shape[:2]
            ws = _native.workspace(H, W, self.device)
            _native.check(_native.lib().mi_isp_linear_batch(
                _native.ptr_array(images), _native.ptr_array(outputs), len(images), H, W, dtype.code,
                self.metrics.data_ptr(), float(gamma), interpolate.transform_code(self.transform), ws.data_ptr(),
                _native.stream_ptr(self.device)))
            return outputs

    ISP.reinhard_kernel = staticmethod(reinhard_kernel)
    ISP.linear_kernel = staticmethod(linear_kernel)
    ISP.dtype = dtype
    ISP.__name__ = name
    ISP.__qualname__ = name
    return ISP


Camera16 = camera_isp("Camera16", types.f16)
Camera32 = camera_isp("Camera32", types.f32)
