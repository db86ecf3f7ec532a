"""Cross-GPU reduction of the rolling metering statistics (one process per GPU, RCCL over xGMI).

Frames are independent, so a sharded batch needs no data-path collective.  The only coupling is
the 9-float metering vector of camera_isp.py:142-166, computed over *all* images of a call in two
dependent data passes.  Each rank runs both passes on its own frames; between them the ranks
exchange 2 floats (raw bounds), after them 8 floats (log bounds, five sums, the pixel count).
Messages are 8..32 bytes -> latency-bound; two all-reduces per exchange keep MIN/MAX and SUM apart.
Every function works on CPU tensors too (gloo), which is how the tests cover world_size 2.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def _active(group) -> bool:
    return group is not None and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


def allreduce_bounds(raw: torch.Tensor, group) -> torch.Tensor:
    """[min, max] over all ranks (camera_isp.py:149-154 across shards)."""
    if not _active(group):
        return raw
    t = torch.stack([-raw[0], raw[1]])
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return torch.stack([-t[0], t[1]])


def blend_bounds(raw: torch.Tensor, prev: torch.Tensor, alpha: float) -> torch.Tensor:
    """camera_isp.py:156-157: lerp(alpha, new, prev) = new + alpha * (prev - new)."""
    return (raw + alpha * (prev[:2] - raw)).to(torch.float32).contiguous()


def allreduce_sums(part: torch.Tensor, group) -> torch.Tensor:
    """part = [log_min, log_max, sum_log, sum_gray, sum_r, sum_g, sum_b, n] -> global values."""
    if not _active(group):
        return part
    mm = torch.stack([-part[0], part[1]])
    sums = part[2:8].clone()
    dist.all_reduce(mm, op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(sums, op=dist.ReduceOp.SUM, group=group)
    return torch.cat([torch.stack([-mm[0], mm[1]]), sums])


def finish_metering(prev: torch.Tensor, b: torch.Tensor, part: torch.Tensor, alpha: float) -> torch.Tensor:
    """camera_isp.py:131-134,164-166: normalise by n, then lerp the 9-vector with the previous
    state (the bounds `b` are already blended once -- the reference blends them twice)."""
    n = part[7]
    v = torch.cat([b[:2], part[0:2], part[2:7] / n]).to(torch.float32)
    return (v + alpha * (prev - v)).to(torch.float32).contiguous()
