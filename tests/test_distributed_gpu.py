"""Two ranks through the HIP kernels (both on cuda:0, fresh child processes, gloo for the few exchanged floats):
config 5 - a Camera16 whose metering is shared over the ranks must give every rank the metrics and outputs of one
unsharded ISP over all frames, over three steps (moving average); config 4 - frames sharded over the ranks with no
shared state give the outputs of a single rank.  RCCL itself needs more than one GPU; what runs here is everything
else of the N > 1 path: mi_isp_metering_bounds / _sums on each rank's shard, the two all-gathers, the combine kernels."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

H, W = 96, 256
STEPS = 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _packed(step, k):
    from tests.util import natural_packed12
    return natural_packed12(np.random.default_rng(1000 + 10 * step + k), H, W, dark=0.02 * k + 0.03 * step)


def _run_isp(frames_of_step, group, dev):
    """3 tonemap_reinhard calls of a Camera16(resize_width=128) on the given frames -> (metrics per step, outputs)."""
    import taichi_image_amd as ti
    isp = ti.Camera16(ti.BayerPattern.RGGB, moving_alpha=0.3, resize_width=128, device=dev, process_group=group)
    metrics, outs = [], []
    for step in range(STEPS):
        imgs = [isp.load_packed12(torch.from_numpy(p).to(dev)) for p in frames_of_step(step)]
        o = isp.tonemap_reinhard(imgs, gamma=0.6)
        metrics.append(isp.metrics.cpu().numpy().copy())
        outs.append([x.cpu().numpy() for x in o])
    return metrics, outs


def _run_group(frames_of_step, group, dev):
    """The same three steps at full resolution through ISP.process_packed12 (the camera-group kernel; sharded: subsample ->
    metering with its two all-gathers -> the persistent launch)."""
    import taichi_image_amd as ti
    isp = ti.Camera16(ti.BayerPattern.RGGB, moving_alpha=0.3, device=dev, process_group=group)
    metrics, outs = [], []
    for step in range(STEPS):
        o = isp.process_packed12([torch.from_numpy(p).to(dev) for p in frames_of_step(step)], gamma=0.6)
        metrics.append(isp.metrics.cpu().numpy().copy())
        outs.append([x.cpu().numpy() for x in o])
    return metrics, outs


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from taichi_image_amd.pipeline import pipeline12_reinhard
    mine = lambda step: [_packed(step, k) for k in range(6)][rank::world]
    metrics, outs = _run_isp(mine, dist.group.WORLD, dev)
    # config 4: this rank's share of independent frames through the stateless chain
    stateless = [pipeline12_reinhard(torch.from_numpy(p).to(dev), whole_frame=False).cpu().numpy() for p in mine(0)]   # two processes share this GPU: not the whole-frame kernel
    # (the frames are small: the resident grids of the two processes fit the chip side by side)
    gm, go = _run_group(mine, dist.group.WORLD, dev)
    q.put((rank, metrics, outs, stateless, gm, go))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_through_the_hip_kernels():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        rank, metrics, outs, stateless, gm, go = q.get(timeout=240)
        got[rank] = (metrics, outs, stateless, gm, go)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # the unsharded run, in this process
    dev = torch.device("cuda", 0)
    from taichi_image_amd.pipeline import pipeline12_reinhard
    metrics, outs = _run_isp(lambda step: [_packed(step, k) for k in range(6)], None, dev)
    for step in range(STEPS):
        for r in range(world):
            m = got[r][0][step]
            assert np.allclose(m, metrics[step], rtol=3e-6, atol=1e-7), (step, r, m, metrics[step])
            for j, o in enumerate(got[r][1][step]):
                want = outs[step][r + world * j]               # rank r holds frames r, r + world, ...
                d = np.abs(o.astype(np.int32) - want.astype(np.int32))
                assert d.max() <= 1 and (d > 0).mean() < 0.01, (step, r, j, d.max())
        assert np.array_equal(got[0][0][step], got[1][0][step])     # identical state on every rank
    # ISP.process_packed12 on a sharded group against the same call on one rank holding all six cameras
    gmetrics, gouts = _run_group(lambda step: [_packed(step, k) for k in range(6)], None, dev)
    for step in range(STEPS):
        for r in range(world):
            m = got[r][3][step]
            assert np.allclose(m, gmetrics[step], rtol=3e-6, atol=1e-7), (step, r, m, gmetrics[step])
            for j, o in enumerate(got[r][4][step]):
                d = np.abs(o.astype(np.int32) - gouts[step][r + world * j].astype(np.int32))
                assert d.max() <= 1 and (d > 0).mean() < 0.01, ("process_packed12", step, r, j, d.max())
        assert np.array_equal(got[0][3][step], got[1][3][step])
    full = [pipeline12_reinhard(torch.from_numpy(_packed(0, k)).to(dev), whole_frame=False).cpu().numpy() for k in range(6)]
    for r in range(world):
        for j, o in enumerate(got[r][2]):
            assert np.array_equal(o.view(np.uint16), full[r + world * j].view(np.uint16)), (r, j)


@pytest.mark.timeout(600)
def test_bench_starts_its_own_ranks_on_one_gpu():
    """`python bench.py --gpus 2` without a launcher: the parent spawns two ranks (gloo rehearsal: both on cuda:0, so the
    multi-pass chain), relays rank 0's line and returns 0."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MI_ISP_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--frames", "2",
                        "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=540)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["value"] > 0
    assert "multi-pass" in line["config"]["chain"]
    # the line explains itself: backend and rank count as the process group reports them, the scaling mode, per-rank times
    assert line["backend"] == "gloo" and line["ranks_seen"] == 2 and line["scaling"] == "weak"
    assert line["config"]["frames_per_step_all_ranks"] == 4 and line["config"]["frames_per_rank_per_step"] == 2
    br = line["us_per_frame_by_rank"]
    assert len(br["per_rank"]) == 2 and 0 < br["min"] <= br["max"]
    # strong scaling: --frames is the total, split over the ranks
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--frames", "4",
                        "--scaling", "strong", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=540)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["scaling"] == "strong" and line["config"]["frames_per_rank_per_step"] == 2
    assert line["config"]["frames_per_step_all_ranks"] == 4 and line["ranks_seen"] == 2
    # config 5 (shared rolling statistics): the collectives' time per step is in the line (gloo: through the host, no events)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--frames", "2",
                        "--workload", "isp-shared-stats"], env=env, capture_output=True, text=True, timeout=540)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["ranks_seen"] == 2 and "collective_us_per_step" in line and "us_per_frame_by_rank" in line
