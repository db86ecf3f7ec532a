"""world_size-2 gloo test of the sharded metering exchange (taichi_image_amd/distributed.py): each rank feeds the
partials of ITS frames (produced here by the oracle, standing in for the HIP metering passes) through the same
two all-gathers the GPU path uses and combines the gathered rows with the arithmetic of the combine kernels
(mi_isp_metering_combine_bounds / _sums, mirrored in NumPy below); every rank must end with the single-process
result over all frames.  The real kernels meet the same check on the GPU in tests/test_distributed_gpu.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import isp_oracle as O

f32 = np.float32


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _frames(n=6):
    rng = np.random.default_rng(77)
    return [(rng.random((48, 64, 3), dtype=np.float32) * (0.5 + 0.1 * i)).astype(np.float16) for i in range(n)]


def combine_bounds(gathered, prev, alpha):
    """metering_combine_bounds_kernel: min / max over the ranks, then camera_isp.py:156-157."""
    lo, hi = f32(gathered[:, 0].min()), f32(gathered[:, 1].max())
    return np.array([lo + f32(alpha) * (prev[0] - lo), hi + f32(alpha) * (prev[1] - hi)], f32)


def combine_sums(gathered, b, prev, alpha):
    """metering_combine_sums_kernel: camera_isp.py:131-134,164-166 over the ranks' rows."""
    lmin, lmax = f32(gathered[:, 0].min()), f32(gathered[:, 1].max())
    sums = gathered[:, 2:7].astype(np.float64).sum(0)
    n = f32(gathered[:, 7].astype(np.float64).sum())
    v = np.array([b[0], b[1], lmin, lmax, *(sums.astype(f32) / n)], f32)
    return (v + f32(alpha) * (prev - v)).astype(f32)


def _worker(rank, world, port, steps, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from taichi_image_amd import distributed as D
    group = dist.group.WORLD
    assert D.active(group) and D.world_size(group) == world
    mine = _frames()[rank::world]
    prev = np.zeros(9, f32)
    alpha = 0.0
    out = []
    for step in range(steps):
        raw = torch.from_numpy(O.metering_partials_bounds(mine))
        gathered = D.all_gather_rows(raw, group).numpy()                       # collective 1
        assert gathered.shape == (world, 2) and np.array_equal(gathered[rank], raw.numpy())
        b = combine_bounds(gathered, prev, alpha)
        part, n = O.metering_partials_sums(mine, b)
        part8 = torch.tensor([*part.astype(np.float32), float(n)], dtype=torch.float32)
        gathered8 = D.all_gather_rows(part8, group).numpy()                    # collective 2
        prev = combine_sums(gathered8, b, prev, alpha)
        out.append(prev.copy())
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
    assert not D.active(None) and D.world_size(None) == 1
    g = D.all_gather_rows(raw, None)
    assert g.shape == (1, 2) and torch.equal(g[0], raw)


@pytest.mark.timeout(300)
def test_bench_spawns_its_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with WORLD_SIZE unset: the parent - before anything touches a GPU - starts two rank
    processes with RANK / WORLD_SIZE / MASTER_* set and returns the worst exit code.  On this GPU-less box each rank
    stops at its "needs a GPU" check (exit code 3), which is the evidence that both were started with the right ranks."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if torch.cuda.is_available():
        pytest.skip("covered by tests/test_distributed_gpu.py::test_bench_starts_its_own_ranks_on_one_gpu")
    env = dict(os.environ, MI_ISP_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2"], env=env,
                       capture_output=True, text=True, timeout=240)
    assert r.returncode == 3, (r.returncode, r.stderr[-1500:])
    for rank in (0, 1):
        assert f"rank {rank}/2" in r.stderr, r.stderr[-1500:]
    # a launcher's environment is taken as it is (no second spawn): WORLD_SIZE=1 with --gpus 2 is the launcher's error
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=dict(env, WORLD_SIZE="1", RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr
