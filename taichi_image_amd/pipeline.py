"""Fused stateless chain of the reference's test/pipeline.py:26-32 (BASELINE config 2):

    decode12(packed, dtype=work, scaled=True) -> bayer_to_rgb -> tonemap_reinhard(dtype=out)

as four data passes (csrc/isp_api.hip: pipeline_frame_cached when the output has the work dtype - one
tile pass that writes the demosaiced image into the output buffer, three elementwise passes on it -
and pipeline_frame, four tile passes over the packed frame, otherwise).  Results are
identical to calling packed.decode12 / bayer.bayer_to_rgb / tonemap.tonemap_reinhard in turn.
"""
from __future__ import annotations

import torch

from . import _native, types
from .bayer import BayerPattern


def _check_packed(packed):
    if not isinstance(packed, torch.Tensor):
        raise TypeError("packed frame must be a torch.Tensor")
    assert packed.ndim == 2 and packed.dtype == torch.uint8, "packed frame must be (H, W*3/2) uint8"
    assert packed.is_cuda and packed.is_contiguous(), "packed frame must be a contiguous device tensor"
    assert packed.shape[1] % 3 == 0, "packed-12 rows must hold whole pixel pairs"
    H, W = packed.shape[0], packed.shape[1] * 2 // 3
    assert H % 2 == 0 and W % 2 == 0, "image must be even size"
    return H, W


def whole_frame_fits(H, W, dtype=types.f16):
    """Can pipeline12_reinhard(..., whole_frame=True) take an H x W frame with this output dtype?"""
    return bool(_native.lib().mi_isp_pipeline12_whole_frame_fits(int(H), int(W), types.as_dtype(dtype).code))


def pipeline12_reinhard(packed, pattern=BayerPattern.RGGB, ids_format=False, correct_colors=None,
                        work_dtype=types.f16, dtype=types.f16, gamma=1.0, intensity=1.0, light_adapt=1.0,
                        color_adapt=0.0, out=None, whole_frame=False):
    """whole_frame=True: the single-launch kernel (csrc/isp_mega.h; f16 work dtype, u8 / u16 / f16 output, frames up
    to 4096 x 3072 on MI355X) instead of the multi-pass chain; same results within the tonemap tolerance."""
    H, W = _check_packed(packed)
    work, odt = types.as_dtype(work_dtype), types.as_dtype(dtype)
    if out is None:
        out = torch.empty((H, W, 3), dtype=odt.torch, device=packed.device)
    ws = _native.workspace(H, W, packed.device)
    if whole_frame:
        assert work.code == types.f16.code, "the whole-frame kernel works in f16"
        _native.check(_native.lib().mi_isp_pipeline12_reinhard_whole_frame(
            packed.data_ptr(), out.data_ptr(), H, W, int(bool(ids_format)), pattern.value,
            _native.ccm_arg(correct_colors), odt.code, float(gamma), float(intensity), float(light_adapt),
            float(color_adapt), ws.data_ptr(), _native.stream_ptr(packed.device)))
        return out
    # an output dtype other than the work dtype: scratch for the work-dtype image between the passes (the
    # caching allocator makes this cheap; without it the library re-derives the image in every pass)
    work_image = None if odt.code == work.code else torch.empty((H, W, 3), dtype=work.torch, device=packed.device)
    _native.check(_native.lib().mi_isp_pipeline12_reinhard(
        packed.data_ptr(), out.data_ptr(), None if work_image is None else work_image.data_ptr(),
        H, W, int(bool(ids_format)), pattern.value, _native.ccm_arg(correct_colors),
        work.code, odt.code, float(gamma), float(intensity), float(light_adapt), float(color_adapt), ws.data_ptr(),
        _native.stream_ptr(packed.device)))
    return out


class BatchPipeline:
    """Independent frames, one frame per HIP stream in flight (BASELINE config 4 on one GPU).

    Owns `n_streams` streams, the per-frame workspaces and the output tensors so that a step is
    one C call issuing 4 launches per frame (7 on the recompute variant) with no allocation on the way."""

    def __init__(self, n_frames, H, W, device, n_streams=2, pattern=BayerPattern.RGGB, ids_format=False,
                 correct_colors=None, work_dtype=types.f16, dtype=types.f16, gamma=1.0, intensity=1.0,
                 light_adapt=1.0, color_adapt=0.0, use_graph=False, whole_frame=False):
        self.n_frames, self.H, self.W, self.device = n_frames, H, W, device
        self.whole_frame = bool(whole_frame)          # every frame through the single-launch kernel (csrc/isp_mega.h)
        if self.whole_frame:
            n_streams = 1                             # two whole-frame grids must never share the chip
        self.work, self.odt = types.as_dtype(work_dtype), types.as_dtype(dtype)
        self.pattern, self.ids = pattern, int(bool(ids_format))
        self.ccm = _native.ccm_arg(correct_colors)
        self.params = (float(gamma), float(intensity), float(light_adapt), float(color_adapt))
        self.streams = [torch.cuda.Stream(device=device) for _ in range(max(1, n_streams))]
        self.stream_ptrs = (_native.c_void_p * len(self.streams))(*[s.cuda_stream for s in self.streams])
        ws_bytes = int(_native.lib().mi_isp_workspace_bytes(H, W))
        self.ws = torch.zeros(ws_bytes * n_frames, dtype=torch.uint8, device=device)
        self.outputs = [torch.empty((H, W, 3), dtype=self.odt.torch, device=device) for _ in range(n_frames)]
        self.out_ptrs = _native.ptr_array(self.outputs)
        # scratch for the work-dtype image of every frame when the outputs have another dtype
        self.work_images = None if self.odt.code == self.work.code else [
            torch.empty((H, W, 3), dtype=self.work.torch, device=device) for _ in range(n_frames)]
        self.work_ptrs = None if self.work_images is None else _native.ptr_array(self.work_images)
        # use_graph: the step (fork to the worker streams, the launches of every frame, join) is captured once
        # into a HIP graph inside the library (mi_isp_pipeline12_graph_create) and replayed while the input
        # tensors keep their addresses (a ring of upload buffers does); replay removes the launch gaps between
        # the dependent kernels
        self.use_graph = bool(use_graph)
        self._graph, self._graph_key = None, None

    def __del__(self):
        g = getattr(self, "_graph", None)
        if g is not None:
            try:
                _native.lib().mi_isp_pipeline12_graph_destroy(g)
            except Exception:
                pass

    def __call__(self, frames, eager=False):
        assert len(frames) == self.n_frames
        for f in frames:
            assert _check_packed(f) == (self.H, self.W)
        if self.use_graph and not eager:
            key = tuple(f.data_ptr() for f in frames)
            if self._graph is None or key != self._graph_key:
                self._issue(frames)                          # warm (lazy initialisation stays out of the capture)
                torch.cuda.synchronize(self.device)
                if self._graph is not None:
                    _native.check(_native.lib().mi_isp_pipeline12_graph_destroy(self._graph))
                    self._graph = None
                handle = _native.c_void_p()
                g_, i_, la_, ca_ = self.params
                with torch.cuda.device(self.device):
                    _native.check(_native.lib().mi_isp_pipeline12_graph_create(
                        _native.ptr_array(frames), self.out_ptrs, self.work_ptrs, self.n_frames, self.H, self.W,
                        self.ids, self.pattern.value, self.ccm, self.work.code, self.odt.code, g_, i_, la_, ca_,
                        self.ws.data_ptr(), len(self.streams), int(self.whole_frame), _native.ctypes.byref(handle)))
                self._graph, self._graph_key, self._graph_inputs = handle, key, list(frames)   # keep the inputs alive
            _native.check(_native.lib().mi_isp_pipeline12_graph_launch(self._graph, _native.stream_ptr(self.device)))
            return self.outputs
        return self._issue(frames)

    def prepare(self, frames):
        """Set-up outside any timed region: library warm-up and (use_graph) the capture for these buffers."""
        self._issue(frames)
        torch.cuda.synchronize(self.device)
        if self.use_graph:
            self(frames)
            torch.cuda.synchronize(self.device)

    def _issue(self, frames):
        with torch.cuda.device(self.device):          # the batch entry points take raw stream arrays: guard here
            return self._issue_on_device(frames)

    def _issue_on_device(self, frames):
        # the frames were produced on the current stream: make the worker streams wait for it
        cur = torch.cuda.current_stream(self.device)
        for s in self.streams:
            s.wait_stream(cur)
        in_ptrs = _native.ptr_array(frames)
        g, i, la, ca = self.params
        if self.whole_frame:
            ws_bytes = int(_native.lib().mi_isp_workspace_bytes(self.H, self.W))
            with torch.cuda.stream(self.streams[0]):
                for k, f in enumerate(frames):
                    _native.check(_native.lib().mi_isp_pipeline12_reinhard_whole_frame(
                        f.data_ptr(), self.outputs[k].data_ptr(), self.H, self.W, self.ids, self.pattern.value, self.ccm,
                        self.odt.code, g, i, la, ca, self.ws.data_ptr() + k * ws_bytes, self.streams[0].cuda_stream))
            cur.wait_stream(self.streams[0])
            return self.outputs
        _native.check(_native.lib().mi_isp_pipeline12_reinhard_batch(
            in_ptrs, self.out_ptrs, self.work_ptrs, self.n_frames, self.H, self.W, self.ids, self.pattern.value, self.ccm,
            self.work.code, self.odt.code, g, i, la, ca, self.ws.data_ptr(), self.stream_ptrs, len(self.streams)))
        for s in self.streams:
            cur.wait_stream(s)
        return self.outputs
