"""Cross-GPU exchange of the rolling metering statistics (one process per GPU, RCCL over xGMI).

Frames are independent, so a sharded batch needs no data-path collective.  The only coupling is the 9-float metering
vector of camera_isp.py:142-166, computed over *all* images of a call in two dependent data passes.  Each rank runs
both passes on its own frames; after the first the ranks all-gather 2 floats (raw bounds), after the second 8 floats
(log bounds, five sums, the pixel count): TWO collectives per call, each followed by one single-thread HIP kernel on
the compute stream that combines the gathered rows and applies the reference's lerps
(mi_isp_metering_combine_bounds / _sums) - no host-side tensor arithmetic in between.
Messages are 8..32 bytes per rank: latency-bound, xGMI bandwidth is irrelevant.

RCCL gathers device tensors in place; under gloo (CPU rehearsal of the control flow, tests) the few floats take a
round trip through the host.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def active(group) -> bool:
    return group is not None and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


def world_size(group) -> int:
    return dist.get_world_size(group) if active(group) else 1


# Measurement hook (bench.py --workload isp-shared-stats): when a list, every RCCL all-gather is bracketed by a pair of
# events on the stream it is enqueued on; the caller reads the pairs behind a synchronisation.  None: no events.
collective_events = None


def all_gather_rows(row: torch.Tensor, group) -> torch.Tensor:
    """(k,) per rank -> (world, k) on every rank; one collective."""
    if not active(group):
        return row.reshape(1, -1)
    world = dist.get_world_size(group)
    if row.is_cuda and dist.get_backend(group) != "gloo":
        out = torch.empty((world, row.numel()), dtype=row.dtype, device=row.device)
        if collective_events is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(row.device))
            dist.all_gather_into_tensor(out, row.contiguous(), group=group)
            e1.record(torch.cuda.current_stream(row.device))
            collective_events.append((e0, e1))
            return out
        dist.all_gather_into_tensor(out, row.contiguous(), group=group)
        return out
    host = row.detach().cpu().contiguous()
    parts = [torch.empty_like(host) for _ in range(world)]
    dist.all_gather(parts, host, group=group)
    return torch.stack(parts).to(row.device)
