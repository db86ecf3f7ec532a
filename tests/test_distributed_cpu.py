"""world_size-2 gloo test of the sharded metering exchange (taichi_image_amd/distributed.py):
each rank feeds the partials of ITS frames (produced here by the oracle, standing in for the HIP
metering passes) through the same collectives the GPU path uses; every rank must end with the
single-process result over all frames."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import isp_oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _frames(n=6):
    rng = np.random.default_rng(77)
    return [(rng.random((48, 64, 3), dtype=np.float32) * (0.5 + 0.1 * i)).astype(np.float16) for i in range(n)]


def _worker(rank, world, port, steps, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from taichi_image_amd import distributed as D
    group = dist.group.WORLD
    frames = _frames()
    mine = frames[rank::world]
    prev = torch.zeros(9, dtype=torch.float32)
    alpha = 0.0
    out = []
    for step in range(steps):
        raw = torch.from_numpy(O.metering_partials_bounds(mine))
        raw = D.allreduce_bounds(raw, group)
        b = D.blend_bounds(raw, prev, alpha)
        part, n = O.metering_partials_sums(mine, b.numpy())
        part8 = torch.tensor([*part.astype(np.float32), float(n)], dtype=torch.float32)
        part8 = D.allreduce_sums(part8, group)
        prev = D.finish_metering(prev, b, part8, alpha)
        out.append(prev.numpy().copy())
        alpha = 0.9
    q.put((rank, np.stack(out)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_sharded_metering_world2():
    world, steps = 2, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, steps, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=100) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    frames = _frames()
    st = O.IspState(moving_alpha=0.1)
    for step in range(steps):
        want = st.update_metering(frames)
        for r in range(world):
            assert np.allclose(results[r][step], want, rtol=2e-6, atol=1e-7), (step, r, results[r][step], want)
    assert np.array_equal(results[0], results[1])      # every rank holds identical state


def test_single_process_passthrough():
    from taichi_image_amd import distributed as D
    raw = torch.tensor([0.1, 0.9])
    assert D.allreduce_bounds(raw, None) is raw
    part = torch.arange(8, dtype=torch.float32)
    assert D.allreduce_sums(part, None) is part
