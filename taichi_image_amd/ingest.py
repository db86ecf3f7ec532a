"""Ingest path: raw frame bytes on the host -> packed frames in HBM, overlapped with the ISP kernels.

The reference feeds its ISP from files (`scripts/tonemap_scan.py:64-87,153-156`): a thread pool reads
each raw file, `torch.frombuffer(...).to(device, non_blocking=True)` uploads it from pageable memory,
and the main thread views the bytes as `(H, W*3/2)` for `isp.load_packed12`.  Same functions here
(`load_raw_bytes`, `load_images_iter`), with the upload done the way the hardware wants it:

* `UploadRing`: N slots of (pinned host buffer, device buffer, copy event, release event).  The
  caller fills `slot.host` (e.g. `file.readinto`), `commit()` starts the H2D copy on the ring's copy
  stream and makes the consuming stream wait for it, `release()` marks the device buffer reusable
  once the kernels issued so far have read it.  Copies of frame n+1 overlap the kernels of frame n.
* PCIe Gen5 x16 moves the 18.9 MB of a 4K packed-12 frame in ~0.35 ms (~36 000 MP/s), well below the
  device rate of the ISP path, so a host-fed deployment is bound by this path.

PyTorch is only the pinned-memory / stream / event provider.
"""
from __future__ import annotations

import os
from concurrent.futures import ThreadPoolExecutor
from typing import Callable, List

import numpy as np
import torch


class _Slot:
    def __init__(self, ring, nbytes, device):
        self.ring = ring
        self.pinned = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
        self.host = self.pinned.numpy()                 # fill this (zero-copy view of the pinned buffer)
        self.device = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self.copied = torch.cuda.Event()
        self.released = torch.cuda.Event()
        self._in_use = False

    def commit(self, nbytes=None, stream=None) -> torch.Tensor:
        """Start the async H2D copy of the first `nbytes` bytes; `stream` (default: the current
        stream) waits for it.  Returns the device bytes (a view of the slot's device buffer)."""
        n = self.pinned.numel() if nbytes is None else int(nbytes)
        stream = stream or torch.cuda.current_stream(self.device.device)
        with torch.cuda.stream(self.ring.copy_stream):
            self.device[:n].copy_(self.pinned[:n], non_blocking=True)
            self.copied.record(self.ring.copy_stream)
        stream.wait_event(self.copied)
        return self.device[:n]

    def release(self, stream=None) -> None:
        """The kernels issued on `stream` so far are the last readers of this slot's device buffer."""
        stream = stream or torch.cuda.current_stream(self.device.device)
        self.released.record(stream)
        self._in_use = False


class UploadRing:
    """Double (N-fold) buffered host -> device upload of fixed-size frames."""

    def __init__(self, n_slots: int, nbytes: int, device: torch.device):
        assert n_slots >= 2, "need at least two slots to overlap copy and compute"
        self.device = device
        self.copy_stream = torch.cuda.Stream(device=device)
        self.slots = [_Slot(self, int(nbytes), device) for _ in range(n_slots)]
        self._next = 0

    def acquire(self) -> _Slot:
        """Next slot in ring order; blocks the host until its previous contents have been consumed
        (kernels done reading the device buffer, hence also the copy out of the pinned buffer)."""
        slot = self.slots[self._next]
        self._next = (self._next + 1) % len(self.slots)
        assert not slot._in_use, "slot acquired again before release(): ring too small for the frames in flight"
        slot.released.synchronize()
        slot._in_use = True
        return slot

    def upload(self, data, stream=None):
        """Convenience: copy `data` (bytes-like or uint8 ndarray) into the next slot and commit it.
        Returns (slot, device bytes); call slot.release() after issuing the kernels that read them."""
        slot = self.acquire()
        src = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data.reshape(-1).view(np.uint8)
        assert src.size <= slot.host.size, f"frame of {src.size} bytes does not fit the {slot.host.size}-byte slots"
        slot.host[:src.size] = src
        return slot, slot.commit(src.size, stream)


_default_rings: dict = {}


def load_raw_bytes(filepath, device: torch.device = torch.device("cuda")) -> torch.Tensor:
    """scripts/tonemap_scan.py:64-69: the raw bytes of a file as a uint8 tensor on `device`, without
    decoding.  Read straight into pinned memory and uploaded on a copy stream; safe to call from worker
    threads (the returned tensor is complete and owns its memory)."""
    size = os.path.getsize(filepath)
    pinned = torch.empty(size, dtype=torch.uint8).pin_memory()
    with open(filepath, "rb") as f:
        got = f.readinto(pinned.numpy())
    assert got == size, f"short read on {filepath}"
    dev = device if device.index is not None else torch.device(device.type, torch.cuda.current_device())
    # the destination is allocated on the caller-side (default) stream's pool, NOT on the copy stream's: the consumer
    # frees it in stream order with its own kernels, and a block owned by a pooled side stream could be handed to a
    # later upload while those kernels still read it
    out = torch.empty(size, dtype=torch.uint8, device=dev)
    stream = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(stream):
        out.copy_(pinned, non_blocking=True)
    stream.synchronize()                                # the worker thread waits, not the caller of the iterator
    return out


def load_images_iter(f: Callable, folders: list, names: List[str]):
    """scripts/tonemap_scan.py:71-87: iterate (name, {folder: tensor}) with one group of look-ahead.
    (The reference's loop submits the first group twice and never yields the last name; every name is
    yielded exactly once here.)"""
    with ThreadPoolExecutor() as executor:
        def add_group(name):
            return {folder: executor.submit(f, folder / name) for folder in folders}
        group = add_group(names[0])
        for i in range(1, len(names) + 1):
            next_group = add_group(names[i]) if i < len(names) else None
            yield names[i - 1], {k: fut.result() for k, fut in group.items()}
            group = next_group
