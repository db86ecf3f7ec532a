#!/usr/bin/env python3
"""Benchmark of the camera-ISP hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic frames: BASELINE config 2,
4096x3072 packed-12 RGGB -> demosaic -> Reinhard tonemap -> f16 RGB (the stateless chain of the
reference's test/pipeline.py:26-32), `--frames` frames per rank per step, one frame per HIP
stream in flight.  Frames are independent, so N ranks shard the batch with no data-path
collective ("scaling": "weak": per-GPU work is fixed).  Inputs are resident in HBM before the
timed region; value = total megapixels (sensor pixels) of all ranks / max-over-ranks wall time.

One JSON line is printed by rank 0; besides the driver's contract keys it carries
  roofline     : the dominant kernel (final map + store pass) timed with events on its stream
  cpu_baseline : the CPU oracle (a port, NOT the reference's Taichi CPU backend, which cannot be
                 installed here) timed on this box's host cores, rank 0 at N=1 only
  kernels_us   : average duration of each of the four data passes
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

H, W = 3072, 4096
MP = H * W / 1e6
BYTES_IN = H * W * 3 // 2            # packed-12 frame
BYTES_OUT_F16 = H * W * 3 * 2        # f16 RGB frame
ALG_BYTES = BYTES_IN + BYTES_OUT_F16  # 94 371 840 B / frame = 7.5 B/px
HBM_PEAK_GBS = 8000.0                # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# the four data passes of the f16 -> f16 pipeline ("cached" variant, csrc/isp_api.hip)
PASS_NAMES = ["pass0: tile_kernel<f16,RGGB,EPI_STORE_MINMAX> (unpack + demosaic -> f16 RGB + bounds)",
              "pass1: rgb_pass_kernel<f16,f16,PM_STATS> (metering sums)",
              "pass2: rgb_pass_kernel<f16,f16,PM_RH_MINMAX> (Reinhard bounds)",
              "pass3: rgb_pass_kernel<f16,f16,PM_RH_STORE> (final map, in place)"]


def time_passes(frame, out, ws_ptr, device, reps=10):
    """Average duration (us) of each data pass, events on the stream the kernels run on."""
    from taichi_image_amd import _native, types
    L = _native.lib()
    stream = torch.cuda.current_stream(device)
    res = []
    for p in range(4):
        def launch():
            _native.check(L.mi_isp_pipeline12_pass(frame.data_ptr(), out.data_ptr(), H, W, 0, 0, None,
                                                   types.f16.code, types.f16.code, 1.0, 1.0, 0.0, p, ws_ptr,
                                                   stream.cuda_stream))
        for _ in range(2):
            launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            launch()
        e1.record(stream)
        e1.synchronize()
        res.append(e0.elapsed_time(e1) * 1e3 / reps)
    return res


def cpu_baseline(packed_frame: np.ndarray):
    """The CPU oracle on a bounded sample of the same workload (rank 0, N=1 only)."""
    ncores = os.cpu_count() or 1
    try:
        from oracle import c_oracle
        if c_oracle.available():
            # the GPU box gives one GPU a share of 16 host cores; stay inside it
            c_oracle.set_threads(min(ncores, 16))
            reps, times = 3, []
            for _ in range(reps):
                t0 = time.perf_counter()
                c_oracle.pipeline12_reinhard(packed_frame)
                times.append(time.perf_counter() - t0)
            t = float(np.median(times))
            return {"value": round(MP / t, 3), "unit": "MP/s", "cores": c_oracle.threads(), "kind": "port",
                    "sample": f"1 full 4096x3072 frame, median of {reps}, C/OpenMP restatement (oracle/isp_oracle.c)"}
    except ImportError:
        pass
    from oracle import isp_oracle as O
    rows = 512
    crop = np.ascontiguousarray(packed_frame[:rows])
    t0 = time.perf_counter()
    O.pipeline12_reinhard(crop)
    t = time.perf_counter() - t0
    return {"value": round(rows * W / 1e6 / t, 3), "unit": "MP/s", "cores": 1, "kind": "port",
            "sample": f"first {rows} rows of one frame ({rows * W / 1e6:.2f} MP), single-threaded NumPy restatement "
                      f"(oracle/isp_oracle.py), host has {ncores} cores"}


def reduce_device(device):
    """Where the max-over-ranks timing tensor lives: the GPU under RCCL, the host in a gloo rehearsal."""
    return torch.device("cpu") if os.environ.get("MI_ISP_BENCH_BACKEND", "nccl") == "gloo" else device


def isp_workload(args, rank, world, device):
    """Configs 3 / 5: the stateful Camera16 chain on `--frames` cameras per rank per step."""
    import torch.distributed as dist
    import taichi_image_amd as ti
    from taichi_image_amd import synthetic
    shared = args.workload == "isp-shared-stats"
    group = dist.group.WORLD if (shared and world > 1) else None
    isp = ti.Camera16(ti.BayerPattern.RGGB, moving_alpha=0.1, resize_width=1920, device=device, process_group=group)
    host = [synthetic.synthetic_packed12((rank * args.frames + i) % 64) for i in range(min(4, args.frames))]
    frames = [torch.from_numpy(host[i % len(host)]).to(device) for i in range(args.frames)]

    def step():
        return isp.tonemap_reinhard([isp.load_packed12(f) for f in frames], gamma=0.6)

    def barrier():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=reduce_device(device))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        out_bytes = 1440 * 1920 * 3
        print(json.dumps({
            "metric": "megapixels/sec end-to-end ISP, 4096x3072 RGGB12; % HBM roofline",
            "value": round(world * args.frames * args.steps * MP / elapsed, 1), "unit": "MP/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("config 5" if shared else "config 3") + ": Camera16(RGGB, resize_width=1920, "
                       "moving_alpha=0.1) load_packed12 x frames + tonemap_reinhard(gamma=0.6) -> u8 1920x1440"
                       + (", metering all-reduced over ranks (RCCL)" if shared else ""),
                       "frames_per_rank_per_step": args.frames},
            "us_per_frame": round(elapsed / (args.frames * args.steps) * 1e6, 2),
            "pipeline_frac_of_hbm_roofline": round((BYTES_IN + out_bytes) * world * args.frames * args.steps / elapsed
                                                   / 1e9 / (HBM_PEAK_GBS * world), 4),
            "roofline": None, "cpu_baseline": None}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=8, help="frames per rank per step")
    ap.add_argument("--streams", type=int, default=2, help="HIP streams per rank (frames in flight)")
    ap.add_argument("--profile-every", type=int, default=5,
                    help="HIP events around the passes of every n-th frame of the timed region")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step as a captured HIP graph (+4-6 %% throughput, DESIGN.md 7).  Not the default: "
                         "kernels inside a graph cannot be timed with events, so the per-pass durations would come "
                         "from the few steps issued launch by launch (--eager-every) and stop agreeing with rocprofv3")
    ap.add_argument("--eager-every", type=int, default=10,
                    help="with --graph: every n-th step of the timed region is issued launch by launch so that "
                         "the per-pass events (--profile-every) can be recorded")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="config2", choices=["config2", "isp", "isp-shared-stats"],
                    help="config2 (default, the BASELINE metric) | isp: Camera16(resize_width=1920) load_packed12 + "
                         "tonemap_reinhard(gamma=0.6), config 3 | isp-shared-stats: the same with the rolling metering "
                         "statistics all-reduced over the ranks (config 5)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    # MI_ISP_BENCH_BACKEND=gloo: rehearsal of the N > 1 control flow on a box with fewer GPUs than
    # ranks (ranks share devices, timing tensors go through the host); the real runs use RCCL
    backend = os.environ.get("MI_ISP_BENCH_BACKEND", "nccl")
    device = torch.device("cuda", local_rank % torch.cuda.device_count() if backend == "gloo" else local_rank)
    torch.cuda.set_device(device)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    from taichi_image_amd import synthetic
    from taichi_image_amd.pipeline import BatchPipeline

    if args.workload != "config2":
        return isp_workload(args, rank, world, device)

    # distinct synthetic frames per rank (seeds 1234 + k); 4 distinct, cycled
    n_distinct = min(4, args.frames)
    host = [synthetic.synthetic_packed12((rank * args.frames + i) % 64) for i in range(n_distinct)]
    frames = [torch.from_numpy(host[i % n_distinct]).to(device) for i in range(args.frames)]
    bp = BatchPipeline(args.frames, H, W, device, n_streams=args.streams, use_graph=args.graph)

    def barrier():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    from taichi_image_amd import _native
    import ctypes
    bp.prepare(frames)                    # set-up (module load, graph capture), not a step
    for _ in range(args.warmup):
        bp(frames)
    barrier()
    # events around every launch of the dominant kernel inside the timed region (rank 0's line)
    _native.check(_native.lib().mi_isp_profile_enable(args.frames * args.steps, args.profile_every))
    t0 = time.perf_counter()
    for step in range(args.steps):
        bp(frames, eager=(not args.graph) or (step % max(1, args.eager_every) == 0))
    barrier()
    elapsed = time.perf_counter() - t0
    live_us, live_n = (ctypes.c_float * 4)(), ctypes.c_int(0)
    _native.check(_native.lib().mi_isp_profile_collect(live_us, ctypes.byref(live_n)))
    _native.check(_native.lib().mi_isp_profile_enable(0, 1))
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=reduce_device(device))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        total_mp = world * args.frames * args.steps * MP
        value = total_mp / elapsed
        ms_per_step = elapsed / args.steps * 1e3
        passes = time_passes(frames[0], bp.outputs[0], bp.ws.data_ptr(), device)
        live = [float(v) for v in live_us]            # in-situ averages over the timed region
        dom = int(np.argmax(live))                    # the dominant kernel = the longest data pass
        dom_us = live[dom]
        # algorithmic bytes per launch of each pass: pass 0 reads the packed frame and writes
        # the f16 image, passes 1-2 read it, pass 3 reads and rewrites it
        pass_bytes = [BYTES_IN + BYTES_OUT_F16, BYTES_OUT_F16, BYTES_OUT_F16, 2 * BYTES_OUT_F16]
        dom_bytes = pass_bytes[dom]
        achieved = dom_bytes / (dom_us * 1e-6) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath))["kernels"][f"pass{dom}"]["hbm_bytes"]
            except Exception:
                traffic = None
        line = {
            "metric": "megapixels/sec end-to-end ISP, 4096x3072 RGGB12; % HBM roofline",
            "value": round(value, 1), "unit": "MP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "config 2: 4096x3072 packed-12 RGGB -> demosaic -> Reinhard tonemap (stateless, "
                                   "gamma 1) -> f16 RGB", "frames_per_rank_per_step": args.frames,
                       "streams_per_rank": args.streams, "work_dtype": "f16",
                       "launch": (f"HIP graph replay of the step; every {args.eager_every}th step launch by launch for the per-pass events"
                                  if args.graph else "launch by launch"), "sharding": f"frames x{world}, no collective"},
            "us_per_frame": round(elapsed / (args.frames * args.steps) * 1e6, 2),
            "pipeline_frac_of_hbm_roofline": round(ALG_BYTES * world * args.frames * args.steps / elapsed / 1e9
                                                   / (HBM_PEAK_GBS * world), 4),
            "roofline": {"bound": "hbm", "kernel": PASS_NAMES[dom],
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_us": round(dom_us, 2),
                         "launches_timed": int(live_n.value), "isolated_launch_us": round(passes[dom], 2),
                         "isolated_frac": round(dom_bytes / (passes[dom] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)},
            "kernels_us_live": {f"pass{k}": round(live[k], 2) for k in range(4)},
            "kernels_us_isolated": {f"pass{k}": round(passes[k], 2) for k in range(4)},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(host[0])
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
