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


class WholeFrameTimeout(RuntimeError):
    """A grid barrier of the whole-frame kernel timed out (something else held compute units: a foreign kernel, another
    process on the GPU).  `.frames`: indices of the frames that were lost and re-issued through the multi-pass chain
    (empty when the fault was noticed through the device's mailbox only and the frames cannot be told any more)."""

    def __init__(self, msg, frames=()):
        super().__init__(msg)
        self.frames = tuple(frames)


def whole_frame_fits(H, W, dtype=types.f16):
    """Can pipeline12_reinhard(..., whole_frame=True) take an H x W frame with this output dtype?"""
    return bool(_native.lib().mi_isp_pipeline12_whole_frame_fits(int(H), int(W), types.as_dtype(dtype).code))


def _auto_whole_frame(H, W, work, odt, ids_format):
    return (work.code == types.f16.code and odt.code in (types.u8.code, types.u16.code, types.f16.code)
            and not ids_format and W % 8 == 0 and whole_frame_fits(H, W, odt))


def _failed_frames(ws, n_frames, H, W, device):
    """Synchronise the current stream and read (and clear) the fault words of n_frames consecutive workspaces."""
    failed = (_native.c_int * n_frames)()
    n_bad = _native.c_int(0)
    _native.check(_native.lib().mi_isp_workspace_check(ws.data_ptr(), n_frames, H, W, failed, _native.ctypes.byref(n_bad),
                                                       _native.stream_ptr(device)))
    return [i for i in range(n_frames) if failed[i]]


def _report(frames_lost, on_timeout, what, later=False):
    msg = (f"{what}: a grid barrier of the whole-frame kernel timed out for frame(s) {list(frames_lost)} - something else "
           "held compute units of this GPU; the frames were re-issued through the multi-pass chain and are valid now"
           + (" (they were NOT while the calls that followed them ran: this is reported at the first call after the fault)"
              if later else ""))
    if on_timeout == "raise":
        raise WholeFrameTimeout(msg, frames_lost)
    import warnings
    warnings.warn(msg, RuntimeWarning, stacklevel=3)


def _multi_pass(packed, out, H, W, ids_format, pattern, ccm, work, odt, params, ws):
    # an output dtype other than the work dtype: scratch for the work-dtype image between the passes (the
    # caching allocator makes this cheap; without it the library re-derives the image in every pass)
    work_image = None if odt.code == work.code else torch.empty((H, W, 3), dtype=work.torch, device=packed.device)
    g, i, la, ca = params
    _native.check(_native.lib().mi_isp_pipeline12_reinhard(
        packed.data_ptr(), out.data_ptr(), None if work_image is None else work_image.data_ptr(),
        H, W, int(bool(ids_format)), pattern.value, ccm, work.code, odt.code, g, i, la, ca, ws.data_ptr(),
        _native.stream_ptr(packed.device)))


# Whole-frame launches of pipeline12_reinhard that have not been looked at yet, per device, oldest first: weak references
# to the caller's tensors (a frame nobody holds any more needs no repair) and what it takes to re-issue the call.
_pending: dict = {}
_PENDING_MAX = 1024
# The mailbox is one word per device, shared by everything that launches the whole-frame kernel there (this function's
# deferred path, every BatchPipeline).  Who finds it set looks at his OWN fault words (sticky, per workspace) and clears the
# mailbox only when the fault was his; otherwise he remembers that the set state belongs to somebody else (and, for the
# deferred path, checks his own launches synchronously until the owner has cleared it).
_not_mine: dict = {}


def _mailbox_set(device, owner):
    """One host read.  False also while the set state is known to be somebody else's."""
    with torch.cuda.device(device):
        v = _native.lib().mi_isp_whole_frame_faults(0)
    if not v:
        _not_mine.pop((device.index or 0, owner), None)
        return False
    return not _not_mine.get((device.index or 0, owner), False)


def _mailbox_settle(device, owner, mine):
    if mine:
        with torch.cuda.device(device):
            _native.lib().mi_isp_whole_frame_faults(1)
    else:
        _not_mine[(device.index or 0, owner)] = True


def check_pending(device=None, on_timeout="fallback"):
    """The deferred check of pipeline12_reinhard's default path, on demand: ONE host read of the device's fault mailbox
    (no synchronisation).  Only when it is set: synchronise, find the calls whose workspace carries the fault word and
    re-issue them through the multi-pass chain on the current stream; then warn (on_timeout="fallback") or raise
    WholeFrameTimeout (="raise").  Returns the number of frames re-issued.  Call it at a point where you synchronise
    anyway and BEFORE consuming outputs when you cannot afford to find out one call later."""
    import weakref  # noqa: F401
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    key = device.index or 0
    if not _mailbox_set(device, "single"):
        return 0
    with torch.cuda.device(device):
        torch.cuda.synchronize(device)
        entries = _pending.pop(key, [])
        bad_ws, redo, gone = {}, [], 0
        for n, (rp, ro, args, ws) in enumerate(entries):
            wkey = ws.data_ptr()
            if wkey not in bad_ws:
                bad_ws[wkey] = bool(_failed_frames(ws, 1, args[0], args[1], device))
            if bad_ws[wkey]:
                packed, out = rp(), ro()
                if packed is None or out is None:
                    gone += 1
                else:
                    redo.append((n, packed, out, args, ws))
        _mailbox_settle(device, "single", any(bad_ws.values()))
        for n, packed, out, args, ws in redo:
            _multi_pass(packed, out, *args, ws)
        if redo or gone:
            _report([n for n, *_ in redo], on_timeout, "pipeline12_reinhard", later=True)
        return len(redo)


def pipeline12_reinhard(packed, pattern=BayerPattern.RGGB, ids_format=False, correct_colors=None,
                        work_dtype=types.f16, dtype=types.f16, gamma=1.0, intensity=1.0, light_adapt=1.0,
                        color_adapt=0.0, out=None, whole_frame=None, check=None, on_timeout="fallback"):
    """decode12 -> bayer_to_rgb -> tonemap_reinhard of one packed frame (test/pipeline.py:26-32 of the reference).

    whole_frame: True = the single-launch kernel (csrc/isp_mega.h; f16 work dtype, u8 / u16 / f16 output, frames up to
    4096 x 3072 on MI355X), False = the multi-pass chain, None (default) = the single-launch kernel whenever it takes the
    frame (never inside a stream capture: a captured launch can be neither ordered nor checked).  Same results within
    the tonemap tolerance.  The single-launch kernel needs the GPU to itself: when a foreign kernel keeps its blocks from
    being resident together its barriers time out and the frame is invalid.
    check: how that is found out.
      "deferred" (the default when the kernel was chosen automatically): NO synchronisation.  Every call first reads the
          device's host-mapped fault mailbox (a plain host read); when an earlier launch has timed out, the calls since
          the last look are checked and the lost frames re-issued through the multi-pass chain - reported (warning, or
          WholeFrameTimeout with on_timeout="raise") at that FOLLOWING call, or at check_pending(), whichever comes first.
          A frame consumed in between was invalid: call check_pending() behind your own synchronisation when that matters.
      True: synchronise right after the launch, re-issue at once (the result is always valid on return).
      False (the default with whole_frame=True): nothing; BatchPipeline.check / mi_isp_whole_frame_faults tell later."""
    import weakref
    H, W = _check_packed(packed)
    work, odt = types.as_dtype(work_dtype), types.as_dtype(dtype)
    if out is None:
        out = torch.empty((H, W, 3), dtype=odt.torch, device=packed.device)
    ws = _native.workspace(H, W, packed.device)
    capturing = torch.cuda.is_current_stream_capturing()
    if check is None:
        check = "deferred" if whole_frame is None else False
    if whole_frame is None:
        whole_frame = not capturing and _auto_whole_frame(H, W, work, odt, ids_format)
    if capturing:
        check = False                                 # (no host-side look is possible at capture time)
    ccm = _native.ccm_arg(correct_colors)
    params = (float(gamma), float(intensity), float(light_adapt), float(color_adapt))
    args = (H, W, ids_format, pattern, ccm, work, odt, params)
    if check == "deferred":
        check_pending(packed.device, on_timeout)      # one host read; repairs and reports what an EARLIER call lost

    if whole_frame:
        assert work.code == types.f16.code, "the whole-frame kernel works in f16"
        _native.check(_native.lib().mi_isp_pipeline12_reinhard_whole_frame(
            packed.data_ptr(), out.data_ptr(), H, W, int(bool(ids_format)), pattern.value, ccm, odt.code, *params,
            ws.data_ptr(), _native.stream_ptr(packed.device)))
        if check == "deferred" and _not_mine.get((packed.device.index or 0, "single"), False):
            check = True                              # the mailbox is held by somebody else's fault: look at our own word
        if check is True:
            if _failed_frames(ws, 1, H, W, packed.device):
                with torch.cuda.device(packed.device):
                    _native.lib().mi_isp_whole_frame_faults(1)
                _multi_pass(packed, out, *args, ws)
                _report([0], on_timeout, "pipeline12_reinhard")
        elif check == "deferred":
            q = _pending.setdefault(packed.device.index or 0, [])
            if len(q) >= _PENDING_MAX:
                del q[:_PENDING_MAX // 2]
            q.append((weakref.ref(packed), weakref.ref(out), args, ws))
        return out
    _multi_pass(packed, out, *args, ws)
    return out


class BatchPipeline:
    """Independent frames, one frame per HIP stream in flight (BASELINE config 4 on one GPU).

    Owns `n_streams` streams, the per-frame workspaces and the output tensors so that a step is
    one C call issuing 4 launches per frame (7 on the recompute variant) with no allocation on the way.

    With the whole-frame kernel (the default for frames it takes) a batch is ONE asynchronous launch, and its outputs are
    valid only if no grid barrier timed out (something else held compute units of the GPU).  Before CONSUMING the outputs
    of a batch either call check() (synchronises, repairs lost frames through the multi-pass chain) or, behind a
    synchronisation of your own, faulted() (one host read of the device's mailbox, no synchronisation: False = every
    whole-frame launch of this process on the device so far was complete).  __call__ looks at the mailbox too and
    repairs the PREVIOUS batch before it issues the next - too late for a caller who has consumed it already."""

    def __init__(self, n_frames, H, W, device, n_streams=2, pattern=BayerPattern.RGGB, ids_format=False,
                 correct_colors=None, work_dtype=types.f16, dtype=types.f16, gamma=1.0, intensity=1.0,
                 light_adapt=1.0, color_adapt=0.0, use_graph=False, whole_frame=None, on_timeout="fallback"):
        self.n_frames, self.H, self.W, self.device = n_frames, H, W, device
        self.work, self.odt = types.as_dtype(work_dtype), types.as_dtype(dtype)
        if whole_frame is None:                       # the single-launch kernel whenever it takes these frames
            with torch.cuda.device(device):
                whole_frame = _auto_whole_frame(H, W, self.work, self.odt, ids_format)
        self.whole_frame = bool(whole_frame)          # the batch through ONE launch of the whole-frame kernel (csrc/isp_mega.h)
        self.on_timeout = on_timeout
        self._last_frames = None
        if self.whole_frame:
            n_streams = 1                             # two whole-frame grids must never share the chip
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
        _not_mine.pop(((self.device.index or 0) if hasattr(self, "device") else 0, id(self)), None)
        g = getattr(self, "_graph", None)
        if g is not None:
            try:
                _native.lib().mi_isp_pipeline12_graph_destroy(g)
            except Exception:
                pass

    def __call__(self, frames, eager=False):
        """Issue the batch (asynchronously).  With the whole-frame kernel: if the device's fault mailbox says that an
        earlier launch timed out, the previous batch is checked and repaired first (see check())."""
        assert len(frames) == self.n_frames
        for f in frames:
            assert _check_packed(f) == (self.H, self.W)
        if self.whole_frame:
            if self._last_frames is not None and _mailbox_set(self.device, id(self)):
                self.check()
            self._last_frames = list(frames)
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

    def faulted(self):
        """Has any whole-frame launch on this device timed out since the mailbox was last cleared?  A host read; meaningful
        for a batch once the caller has synchronised with it."""
        if not self.whole_frame:
            return False
        with torch.cuda.device(self.device):
            return bool(_native.lib().mi_isp_whole_frame_faults(0))

    def check(self, frames=None):
        """Synchronise and make sure every output of the last batch is valid: frames whose fault word is set (a barrier
        of the whole-frame kernel timed out) are re-issued through the multi-pass chain; returns their indices
        (on_timeout="raise": raises WholeFrameTimeout after the repair).  No-op for the multi-pass chain."""
        frames = self._last_frames if frames is None else frames
        with torch.cuda.device(self.device):
            torch.cuda.synchronize(self.device)
            if not self.whole_frame:
                return []
            lost = _failed_frames(self.ws, self.n_frames, self.H, self.W, self.device)
            if _native.lib().mi_isp_whole_frame_faults(0):
                _mailbox_settle(self.device, id(self), bool(lost))
            if lost:
                assert frames is not None, "check(): pass the frames of the batch that was lost"
                g, i, la, ca = self.params
                ws_bytes = int(_native.lib().mi_isp_workspace_bytes(self.H, self.W))
                for k in lost:
                    _native.check(_native.lib().mi_isp_pipeline12_reinhard(
                        frames[k].data_ptr(), self.outputs[k].data_ptr(), None, self.H, self.W, self.ids, self.pattern.value,
                        self.ccm, self.work.code, self.odt.code, g, i, la, ca, self.ws.data_ptr() + k * ws_bytes,
                        _native.stream_ptr(self.device)))
                torch.cuda.synchronize(self.device)
                _report(lost, self.on_timeout, "BatchPipeline")
        return lost

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
            # one launch for the whole batch: the resident grid walks through the frames (csrc/isp_mega.h)
            _native.check(_native.lib().mi_isp_pipeline12_reinhard_whole_frame_batch(
                in_ptrs, self.out_ptrs, self.n_frames, self.H, self.W, self.ids, self.pattern.value, self.ccm,
                self.odt.code, g, i, la, ca, self.ws.data_ptr(), self.streams[0].cuda_stream))
            cur.wait_stream(self.streams[0])
            return self.outputs
        _native.check(_native.lib().mi_isp_pipeline12_reinhard_batch(
            in_ptrs, self.out_ptrs, self.work_ptrs, self.n_frames, self.H, self.W, self.ids, self.pattern.value, self.ccm,
            self.work.code, self.odt.code, g, i, la, ca, self.ws.data_ptr(), self.stream_ptrs, len(self.streams)))
        for s in self.streams:
            cur.wait_stream(s)
        return self.outputs
