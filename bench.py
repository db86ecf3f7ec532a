#!/usr/bin/env python3
"""Benchmark of the camera-ISP hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic frames: BASELINE config 2,
4096x3072 packed-12 RGGB -> demosaic -> Reinhard tonemap -> f16 RGB (the stateless chain of the
reference's test/pipeline.py:26-32), `--frames` frames per rank per step (default 64, BASELINE's
batch).  Headline chain: the whole-frame kernel (csrc/isp_mega.h), ONE launch per step - the resident
grid walks through the 64 frames.  (--chain multi-pass: the streaming chain, one frame per HIP stream
in flight, the step replayed as a HIP graph captured inside the library.)  Frames are independent, so
N ranks shard the batch with no data-path collective.  --scaling weak (default; "scaling": "weak"): every rank
processes --frames frames per step, per-GPU work is fixed; --scaling strong: --frames frames per step in TOTAL,
--frames / N per rank - BASELINE config 4's literal shape (64 frames over 8 GPUs = 8 frames per launch, so the
~14 us a launch costs weigh 1.7 us per frame instead of 0.2).  The line says which one ran, which backend the process
group has, how many ranks IT counts, and the fastest / slowest rank's own time per frame.
Inputs are resident in HBM before the timed region; value = total megapixels (sensor pixels) of all
ranks / max-over-ranks wall time.

Started without a launcher (`python bench.py --gpus N`, WORLD_SIZE unset) and N > 1, the process - before it
touches a GPU - spawns N rank processes of itself (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set), relays
rank 0's JSON line and exits with the worst return code.

One JSON line is printed by rank 0; besides the driver's contract keys it carries
  roofline        : the dominant kernel (the pass that reads the packed frame and writes the output: final map +
                    store) timed with HIP events on the stream it runs on, over launch-by-launch steps interleaved in
                    the timed region
  cpu_baseline    : the CPU oracle (a port, NOT the reference's Taichi CPU backend, which cannot be
                    installed here) timed on this box's host cores, rank 0 at N=1 only
  kernels_us_*    : average duration of each data pass, in situ and isolated
  other_workloads : (N=1) driver-run numbers of the other single-GPU configurations - config 3, config 2 with u8
                    output, config 2 on frames whose bounds are not (0, 1) (no data-dependent shortcut), and
                    config 2 through the single-launch whole-frame kernel
"""
from __future__ import annotations

import argparse
import ctypes
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
METRIC = "megapixels/sec end-to-end ISP, 4096x3072 RGGB12; % HBM roofline"
# the data passes of the f16 -> f16 chain on the streaming kernels (csrc/isp_stream.h, csrc/isp_api.hip:
# pipeline_frame_stream): every pass re-derives the demosaiced image from the packed frame
PASS_NAMES = ["pass0: stream_kernel<f16,RGGB,S_BOUNDS> (unpack + demosaic -> bounds + speculative statistics)",
              "pass1: stream_kernel<f16,RGGB,S_STATS> (statistics for bounds other than (0, 1); returns at once otherwise)",
              "pass2: stream_kernel<f16,RGGB,S_RH_MINMAX> (unpack + demosaic + Reinhard -> bounds of the mapped image)",
              "pass3: stream_kernel<f16,RGGB,S_RH_STORE> (unpack + demosaic + Reinhard + final map -> f16 RGB)"]
# algorithmic bytes per launch: passes 0-2 read the packed frame, pass 3 reads it and writes the output
PASS_BYTES = [BYTES_IN, BYTES_IN, BYTES_IN, BYTES_IN + BYTES_OUT_F16]


def time_passes(frame, out, ws_ptr, device, reps=20):
    """Average duration (us) of each data pass in isolation, events on the stream the kernels run on."""
    from taichi_image_amd import _native, types
    L = _native.lib()
    stream = torch.cuda.current_stream(device)
    res = []
    for p in range(4):
        def launch():
            _native.check(L.mi_isp_pipeline12_pass(frame.data_ptr(), out.data_ptr(), H, W, 0, 0, None,
                                                   types.f16.code, types.f16.code, 1.0, 1.0, 0.0, p, ws_ptr,
                                                   stream.cuda_stream))
        for _ in range(3):
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
    """The CPU oracle on a bounded sample of the same workload (rank 0, N=1 only): median of 10 after 3 warm-ups."""
    ncores = os.cpu_count() or 1
    try:
        from oracle import c_oracle
        if c_oracle.available():
            # the GPU box gives one GPU a share of 16 host cores; stay inside it
            c_oracle.set_threads(min(ncores, 16))
            warm, reps, times = 3, 10, []
            for i in range(warm + reps):
                t0 = time.perf_counter()
                c_oracle.pipeline12_reinhard(packed_frame)
                if i >= warm:
                    times.append(time.perf_counter() - t0)
            t = float(np.median(times))
            return {"value": round(MP / t, 3), "unit": "MP/s", "cores": c_oracle.threads(), "kind": "port",
                    "sample": f"1 full 4096x3072 frame per run, median of {reps} runs after {warm} warm-ups, "
                              f"C/OpenMP restatement (oracle/isp_oracle.c), host has {ncores} hardware threads"}
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


def host_frames(first, count, distinct):
    """`distinct` different synthetic frames (seeds 1234 + index), generated on a thread pool (3 s of numpy
    each), cycled over `count` slots.  Returns (list of the distinct host frames, index of every slot)."""
    from concurrent.futures import ThreadPoolExecutor
    from taichi_image_amd import synthetic
    distinct = max(1, min(distinct, count))
    # (the ranks of a node generate their frames at the same time: each takes its share of the host's cores)
    share = max(2, (os.cpu_count() or 4) // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))))
    with ThreadPoolExecutor(min(16, share, distinct)) as ex:
        host = list(ex.map(lambda i: synthetic.synthetic_packed12((first + i) % 64), range(distinct)))
    return host, [i % distinct for i in range(count)]


def rank_stats(elapsed_local, frames_local, steps, world, device):
    """Every rank's own time for its share (local synchronisation, before the ranks' barrier): us per frame of the
    fastest and the slowest rank - what tells a straggler from a uniformly slow run in the driver's 8-GPU line."""
    us = elapsed_local / max(1, frames_local * steps) * 1e6
    if world == 1:
        return {"min": round(us, 2), "max": round(us, 2)}
    import torch.distributed as dist
    t = torch.tensor([us], dtype=torch.float64, device=reduce_device(device))
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t)
    vals = [float(p.item()) for p in parts]
    return {"min": round(min(vals), 2), "max": round(max(vals), 2), "per_rank": [round(v, 2) for v in vals]}


def run_identity(world, backend):
    """Who took part: the backend of the process group and the number of ranks IT reports (not the --gpus argument)."""
    if world > 1:
        import torch.distributed as dist
        return {"backend": "rccl (torch.distributed 'nccl')" if dist.get_backend() == "nccl" else dist.get_backend(),
                "ranks_seen": dist.get_world_size()}
    return {"backend": "none (single process)", "ranks_seen": 1}


def reduce_device(device):
    """Where the max-over-ranks timing tensor lives: the GPU under RCCL, the host in a gloo rehearsal."""
    return torch.device("cpu") if os.environ.get("MI_ISP_BENCH_BACKEND", "nccl") == "gloo" else device


def timed(fn, steps, warmup, device, barrier=None, local=None):
    """`steps` calls of fn bracketed by synchronisation (and the ranks' barrier); seconds.  local: a list that receives this
    rank's own time (its device synchronised, before the barrier)."""
    sync = barrier or (lambda: torch.cuda.synchronize(device))
    for _ in range(warmup):
        fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    if local is not None:
        torch.cuda.synchronize(device)
        local.append(time.perf_counter() - t0)
    sync()
    return time.perf_counter() - t0


def isp_step_fn(frames_dev, device, process_group=None, full_res=False):
    import taichi_image_amd as ti
    if full_res:
        # the reference bench's step (bench/camera_isp.py:23-27) at full resolution, as one call (csrc/isp_mega_cam.h)
        isp = ti.Camera16(ti.BayerPattern.RGGB, moving_alpha=0.1, device=device, process_group=process_group)
        return lambda: isp.process_packed12(frames_dev, gamma=0.6)
    isp = ti.Camera16(ti.BayerPattern.RGGB, moving_alpha=0.1, resize_width=1920, device=device,
                      process_group=process_group)
    return lambda: isp.tonemap_reinhard([isp.load_packed12(f) for f in frames_dev], gamma=0.6)


def isp_workload(args, rank, world, device):
    """Configs 3 / 5: the stateful Camera16 chain on `--frames` cameras per rank per step."""
    import torch.distributed as dist
    from taichi_image_amd import synthetic
    shared = args.workload.endswith("shared-stats")
    full_res = args.workload.startswith("camera-group")
    if full_res and world > 1 and world > torch.cuda.device_count():
        # (ranks that share a GPU - the gloo rehearsal - would put two resident grids on one chip)
        sys.exit("bench.py: the camera-group workloads need one GPU per rank")
    group = dist.group.WORLD if (shared and world > 1) else None
    host, slot = host_frames(rank * args.frames, args.frames, min(4, args.frames))
    frames = [torch.from_numpy(host[slot[i]]).to(device) for i in range(args.frames)]
    step = isp_step_fn(frames, device, group, full_res)

    def barrier():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    from taichi_image_amd import distributed as tdist
    local = []
    elapsed = timed(step, args.steps, args.warmup, device, barrier, local)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=reduce_device(device))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    per_rank = rank_stats(local[0], args.frames, args.steps, world, device)
    # what the two collectives of a step cost (config 5): HIP events around every all-gather of a few extra steps BEHIND
    # the timed region (events between launches are not free)
    collective_us = None
    if group is not None:
        tdist.collective_events = []
        for _ in range(8):
            step()
        torch.cuda.synchronize(device)
        ev = tdist.collective_events
        tdist.collective_events = None
        if ev:
            collective_us = round(sum(e0.elapsed_time(e1) for e0, e1 in ev) * 1e3 / 8, 2)
    if rank == 0:
        out_bytes = H * W * 3 if full_res else 1440 * 1920 * 3
        print(json.dumps({
            **run_identity(world, os.environ.get("MI_ISP_BENCH_BACKEND", "nccl")),
            "us_per_frame_by_rank": per_rank,
            "collective_us_per_step": collective_us,
            "collective_note": ("two all-gathers per update_metering (2 + 8 floats per rank), HIP events on the compute stream, "
                                "average over 8 steps behind the timed region" if collective_us is not None else None),
            "metric": METRIC,
            "value": round(world * args.frames * args.steps * MP / elapsed, 1), "unit": "MP/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (("config 5 at full resolution" if shared else "the reference's bench step (bench/camera_isp.py:19-28)")
                                    + ": Camera16(RGGB, moving_alpha=0.1).process_packed12(frames, gamma=0.6) -> u8 4096x3072, "
                                    "one call per step (subsample from the packed frames -> metering -> one persistent launch)"
                                    if full_res else
                                    ("config 5" if shared else "config 3") + ": Camera16(RGGB, resize_width=1920, "
                                    "moving_alpha=0.1) load_packed12 x frames + tonemap_reinhard(gamma=0.6) -> u8 1920x1440")
                       + (", metering all-reduced over ranks (RCCL)" if shared else ""),
                       "frames_per_rank_per_step": args.frames},
            "us_per_frame": round(elapsed / (args.frames * args.steps) * 1e6, 2),
            "pipeline_frac_of_hbm_roofline": round((BYTES_IN + out_bytes) * world * args.frames * args.steps / elapsed
                                                   / 1e9 / (HBM_PEAK_GBS * world), 4),
            "roofline": None, "cpu_baseline": None}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def ti_mod():
    import taichi_image_amd
    return taichi_image_amd


def other_workloads(frames, host, device, frames_per_step):
    """Short driver-visible runs of the other single-GPU configurations (rank 0, N=1).  Every entry: three timed runs,
    the median reported and all three listed (no pauses between the runs: round 2 slept 0.3 s in front of each)."""
    from taichi_image_amd import types
    from taichi_image_amd.pipeline import BatchPipeline, whole_frame_fits
    from taichi_image_amd.synthetic import pack12
    res = {}

    def entry(name, us_runs, alg_bytes, n_frames, steps, **more):
        us = float(np.median(us_runs))
        res[name] = {"MP_per_s": round(MP / us * 1e6, 1), "us_per_frame": round(us, 2),
                     "us_per_frame_runs": [round(u, 2) for u in us_runs],
                     "frac_of_hbm_roofline": round(alg_bytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                     "algorithmic_bytes_per_frame": alg_bytes, "frames_per_step": n_frames, "steps": steps, **more}

    def run_batch(name, fr, alg_bytes, steps, **kw):
        bp = BatchPipeline(len(fr), H, W, device, use_graph=True, **kw)
        bp.prepare(fr)
        runs = [timed(lambda: bp(fr), steps, 3, device) / (steps * len(fr)) * 1e6 for _ in range(3)]
        lost = bp.check(fr)
        entry(name, runs, alg_bytes, len(fr), steps, whole_frame_faults=len(lost),
              chain="whole-frame kernel, one launch per step" if bp.whole_frame else "multi-pass chain")
        del bp

    def rescale(p):
        b = p.reshape(H, -1, 3).astype(np.uint32)
        v = np.stack([b[..., 0] | ((b[..., 1] & 0xF) << 8), (b[..., 1] >> 4) | (b[..., 2] << 4)], -1).reshape(H, W)
        v = np.rint(v * 0.7 + 0.1 * 4095).astype(np.uint16)
        return pack12(v)
    # frames whose demosaiced bounds are NOT (0, 1): the statistics phase / pass runs (no data-dependent
    # shortcut); scene scaled into [0.1, 0.8]
    nu_dev = [torch.from_numpy(rescale(host[i])).to(device) for i in range(min(4, len(host)))]
    nonunit = [nu_dev[i % len(nu_dev)] for i in range(len(frames))]
    wf_steps = max(4, 1200 // max(1, len(frames)))         # ~60 ms per run at 64 frames per step
    # config 2 with the u8 output of the ISP semantics (4.5 B/px), through the headline chain
    run_batch("config2_u8_out", frames, BYTES_IN + H * W * 3, wf_steps, dtype=types.u8)
    if whole_frame_fits(H, W, types.f16):
        run_batch("config2_whole_frame_kernel_bounds_not_unit", nonunit, ALG_BYTES, wf_steps, whole_frame=True)
    # config 2 through the other chain (the headline is the whole-frame kernel when the frame fits it)
    run_batch("config2_multi_pass_chain_2_streams", frames[:8], ALG_BYTES, 100, n_streams=2, whole_frame=False)
    run_batch("config2_multi_pass_chain_bounds_not_unit", nonunit[:8], ALG_BYTES, 100, n_streams=2, whole_frame=False)
    # config 3: Camera16(resize_width=1920): load_packed12 x frames + tonemap_reinhard(gamma=0.6) -> u8 1920x1440
    step = isp_step_fn(frames[:6], device)
    steps = 100
    runs = [timed(step, steps, 5, device) / (steps * 6) * 1e6 for _ in range(3)]
    entry("config3_camera16_resize1920", runs, BYTES_IN + 1440 * 1920 * 3, 6, steps)
    # the same step with the group's cameras loaded in one launch (ISP.load_packed12_batch, an extension of the call surface:
    # the reference's load_packed12 is a call per camera) - same outputs, bit for bit
    isp_b = ti_mod().Camera16(ti_mod().BayerPattern.RGGB, moving_alpha=0.1, resize_width=1920, device=device)
    step_b = lambda: isp_b.tonemap_reinhard(isp_b.load_packed12_batch(frames[:6]), gamma=0.6)
    runs = [timed(step_b, steps, 5, device) / (steps * 6) * 1e6 for _ in range(3)]
    entry("config3_camera16_resize1920_batched_load", runs, BYTES_IN + 1440 * 1920 * 3, 6, steps,
          note="extension: ISP.load_packed12_batch (one launch for the six cameras' loads)")
    # ... and without the reference's in-place write of the mapped values over the images (camera_isp.py:211): same u8 outputs
    step_k = lambda: isp_b.tonemap_reinhard(isp_b.load_packed12_batch(frames[:6]), gamma=0.6, write_back=False)
    runs = [timed(step_k, steps, 5, device) / (steps * 6) * 1e6 for _ in range(3)]
    entry("config3_camera16_resize1920_batched_load_no_write_back", runs, BYTES_IN + 1440 * 1920 * 3, 6, steps,
          note="extensions: load_packed12_batch + tonemap_reinhard(write_back=False): the images are NOT overwritten with p "
               "(the reference overwrites them); same u8 outputs bit for bit.  At this image size the second evaluation of "
               "Reinhard costs more than the bytes it saves: slower than the line above; it pays at full resolution (below)")
    # config 3b: the nominal 1920x1080 through the resize primitive's per-axis scale (interpolate.py:83)
    # on the demosaiced 4K f16 image: load_packed12 at full size, resize_bilinear(scale=(0.3515625, 0.46875)) -> f16 / u8
    import taichi_image_amd as ti
    from taichi_image_amd import interpolate
    isp_full = ti.Camera16(ti.BayerPattern.RGGB, device=device)
    for odt, nbytes, tag in ((types.f16, 31315968, "f16"), (types.u8, 25095168, "u8")):
        def step3b():
            for f in frames[:4]:
                interpolate.resize_bilinear(isp_full.load_packed12(f), (1920, 1080), scale=(0.3515625, 0.46875), dtype=odt)
        runs = [timed(step3b, 50, 3, device) / (50 * 4) * 1e6 for _ in range(3)]
        entry(f"config3b_resize_1920x1080_{tag}", runs, nbytes, 4, 50,
              note="load_packed12 (full-size f16 RGB materialised, as the reference does) + resize_bilinear with the per-axis scale")
    # the reference's own bench workload (bench/camera_isp.py:19-45): 6 cameras at full resolution, Camera16 + reinhard
    isp6 = ti.Camera16(ti.BayerPattern.RGGB, moving_alpha=0.1, device=device)
    def step6():
        isp6.tonemap_reinhard([isp6.load_packed12(f) for f in frames[:6]], gamma=0.6)
    runs = [timed(step6, 40, 3, device) / (40 * 6) * 1e6 for _ in range(3)]
    entry("reference_bench_6_cameras_full_resolution", runs, BYTES_IN + H * W * 3, 6, 40)
    def step6b():
        isp6.tonemap_reinhard(isp6.load_packed12_batch(frames[:6]), gamma=0.6)
    runs = [timed(step6b, 40, 3, device) / (40 * 6) * 1e6 for _ in range(3)]
    entry("reference_bench_6_cameras_full_resolution_batched_load", runs, BYTES_IN + H * W * 3, 6, 40,
          note="extension: ISP.load_packed12_batch")
    def step6k():
        isp6.tonemap_reinhard(isp6.load_packed12_batch(frames[:6]), gamma=0.6, write_back=False)
    runs = [timed(step6k, 40, 3, device) / (40 * 6) * 1e6 for _ in range(3)]
    entry("reference_bench_6_cameras_full_resolution_batched_load_no_write_back", runs, BYTES_IN + H * W * 3, 6, 40,
          note="extensions: load_packed12_batch + tonemap_reinhard(write_back=False) (images not overwritten with p; same u8 outputs)")
    # ... and as ONE call per step: the bench's Processor (bench/camera_isp.py:23-27) drops the loaded images, so they need
    # not exist - ISP.process_packed12 on the camera-group kernel (csrc/isp_mega_cam.h)
    isp6c = ti.Camera16(ti.BayerPattern.RGGB, moving_alpha=0.1, device=device)
    def step6c():
        isp6c.process_packed12(frames[:6], gamma=0.6)
    runs = [timed(step6c, 40, 3, device) / (40 * 6) * 1e6 for _ in range(3)]
    entry("reference_bench_6_cameras_full_resolution_one_call", runs, BYTES_IN + H * W * 3, 6, 40,
          note="extension: ISP.process_packed12 = tonemap_reinhard([load_packed12(f) ...]) in one call, same u8 outputs and "
               "metering state bit for bit; the loaded images (which the reference's bench drops) are never written: "
               "subsample from the packed frames -> metering -> one persistent launch from packed bytes to u8")
    def step6ck():
        isp6c.process_packed12(frames[:6], gamma=0.6, keep_images=True)
    runs = [timed(step6ck, 40, 3, device) / (40 * 6) * 1e6 for _ in range(3)]
    entry("reference_bench_6_cameras_full_resolution_one_call_images_kept", runs, BYTES_IN + H * W * 3 + H * W * 6, 6, 40,
          note="the same with keep_images=True: the images the reference leaves behind (p, camera_isp.py:211) are stored too")
    return res


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: N rank processes of this script, before anything touches a GPU."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # rank 0's line is read by a thread; the ranks are watched: when one ends badly (no such GPU, a failed build ...) the
    # others - which would wait in the rendezvous or in a barrier for minutes - are ended too (these exact processes)
    import threading
    import time
    out = []
    reader = threading.Thread(target=lambda: out.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = 0
    while any(p.poll() is None for p in procs):
        bad = [p.returncode for p in procs if p.poll() not in (None, 0)]
        if bad:
            failed = abs(bad[0])
            sys.stderr.write(f"bench.py: a rank ended with code {bad[0]}; ending the other ranks\n")
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            deadline = time.time() + 10
            for p in procs:
                try:
                    p.wait(timeout=max(0.1, deadline - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        time.sleep(0.2)
    rcs = [p.wait() for p in procs]
    reader.join(timeout=5)
    sys.stdout.write("".join(o for o in out if o))
    sys.stdout.flush()
    return failed or max((abs(rc) for rc in rcs), default=0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="steps of the timed region (default: 160 for config 2 = 64 frames each, ~0.5 s; 200 for the ISP workloads)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps (default 5)")
    ap.add_argument("--frames", type=int, default=None,
                    help="frames per rank per step (default: 64 for config 2, BASELINE's batch; 8 for the ISP workloads)")
    ap.add_argument("--streams", type=int, default=2, help="HIP streams per rank (frames in flight)")
    ap.add_argument("--profile-every", type=int, default=2,
                    help="HIP events around the passes of every n-th frame of the launch-by-launch steps")
    ap.add_argument("--no-graph", action="store_true",
                    help="issue every step launch by launch instead of replaying the captured HIP graph")
    ap.add_argument("--eager-every", type=int, default=None,
                    help="every n-th step of the timed region is issued launch by launch so that the per-pass "
                         "events (--profile-every) can be recorded (events inside a graph cannot be timed); default: "
                         "1 for the whole-frame chain (one 3 ms launch per step: nothing to gain from a graph), 25 for "
                         "the multi-pass chain")
    ap.add_argument("--chain", default="auto", choices=["auto", "whole-frame", "multi-pass"],
                    help="config 2 through the single-launch whole-frame kernel (csrc/isp_mega.h; frames one after the "
                         "other) or through the multi-pass streaming chain (frames on --streams streams); auto = the "
                         "whole-frame kernel when the frame fits it")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak (default): --frames frames per RANK per step, the batch grows with N; strong: --frames frames per "
                         "step in TOTAL, --frames / N per rank (BASELINE config 4's literal shape: 64 frames over 8 GPUs)")
    ap.add_argument("--distinct", type=int, default=None,
                    help="distinct synthetic frames per rank (default: every frame of the step its own scene, seeds 1234 + k)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-isolated", action="store_true",
                    help="skip the single-frame launches behind the timed region (profiling: every launch the profiler sees "
                         "is then a headline launch)")
    ap.add_argument("--no-other-workloads", action="store_true")
    ap.add_argument("--workload", default="config2", choices=["config2", "isp", "isp-shared-stats", "camera-group", "camera-group-shared-stats"],
                    help="config2 (default, the BASELINE metric) | isp: Camera16(resize_width=1920) load_packed12 + "
                         "tonemap_reinhard(gamma=0.6), config 3 | isp-shared-stats: the same with the rolling metering "
                         "statistics all-reduced over the ranks (config 5) | camera-group[-shared-stats]: full-resolution "
                         "Camera16.process_packed12 (the reference bench's step as one call), optionally with the shared metering")
    args = ap.parse_args()
    isp = args.workload != "config2"
    if args.steps is None:
        args.steps = 200 if isp else 160
    if args.warmup is None:
        args.warmup = 5
    if args.frames is None:
        args.frames = 8 if isp else 64

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    frames_total = args.frames * world if args.scaling == "weak" else args.frames
    if args.scaling == "strong":
        assert args.frames % world == 0, f"--scaling strong: --frames {args.frames} must be a multiple of --gpus {world}"
        args.frames //= world                        # from here on: frames per rank and step
    if not torch.cuda.is_available():
        print(f"bench.py rank {rank}/{world}: needs a GPU (there is no CPU fallback)", file=sys.stderr, flush=True)
        sys.exit(3)
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

    if isp:
        return isp_workload(args, rank, world, device)

    # synthetic frames of this rank (frame k of the batch is scene k, seed 1234 + k): every frame its own
    # scene unless --distinct asks for fewer (cycled)
    n_distinct = min(args.frames, args.distinct if args.distinct else args.frames)
    host, slot = host_frames(rank * args.frames, args.frames, n_distinct)
    dev_distinct = [torch.from_numpy(h).to(device) for h in host]
    frames = [dev_distinct[slot[i]] if n_distinct < args.frames else dev_distinct[i] for i in range(args.frames)]
    use_graph = not args.no_graph
    from taichi_image_amd import types as _types
    from taichi_image_amd.pipeline import whole_frame_fits
    # ranks that share a device (the gloo rehearsal on a box with fewer GPUs than ranks) cannot use the whole-frame
    # kernel: it needs the chip to itself and only orders launches inside ONE process
    shared_device = world > 1 and backend == "gloo" and world > torch.cuda.device_count()
    whole = args.chain == "whole-frame" or (args.chain == "auto" and whole_frame_fits(H, W, _types.f16) and not shared_device)
    if args.eager_every is None:
        args.eager_every = 1 if whole else 25
    if whole:
        args.profile_every = 1                     # every launch of the timed region carries its events
    bp = BatchPipeline(args.frames, H, W, device, n_streams=args.streams, use_graph=use_graph, whole_frame=whole)

    def barrier():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    from taichi_image_amd import _native
    bp.prepare(frames)                    # set-up (module load, graph capture), not a step
    for _ in range(args.warmup):
        bp(frames)
    barrier()
    # events around every data pass of the launch-by-launch steps inside the timed region (rank 0's line)
    n_eager = (args.steps + args.eager_every - 1) // max(1, args.eager_every) if use_graph else args.steps
    _native.check(_native.lib().mi_isp_profile_enable((1 if whole else args.frames) * n_eager, args.profile_every))
    t0 = time.perf_counter()
    for step in range(args.steps):
        bp(frames, eager=(not use_graph) or (step % max(1, args.eager_every) == 0))
    torch.cuda.synchronize(device)
    elapsed_local = time.perf_counter() - t0          # this rank's own time, before it waits for the others
    barrier()
    elapsed = time.perf_counter() - t0
    live_us, live_n = (ctypes.c_float * 4)(), ctypes.c_int(0)
    _native.check(_native.lib().mi_isp_profile_collect(live_us, ctypes.byref(live_n)))
    _native.check(_native.lib().mi_isp_profile_enable(0, 1))
    faults = len(bp.check(frames)) if whole else 0          # after the timed region: every output valid, or repaired and counted
    per_rank = rank_stats(elapsed_local, args.frames, args.steps, world, device)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=reduce_device(device))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        total_mp = frames_total * args.steps * MP
        value = total_mp / elapsed
        ms_per_step = elapsed / args.steps * 1e3
        live = [float(v) for v in live_us]            # in-situ averages over the timed region
        if whole:
            # one kernel per step: it IS the dominant kernel (reads the packed frames, writes the outputs)
            from taichi_image_amd.pipeline import pipeline12_reinhard
            iso = 0.0 if args.no_isolated else timed(
                lambda: pipeline12_reinhard(frames[0], out=bp.outputs[0], whole_frame=True, check=False), 100, 10, device) / 100 * 1e6
            passes = [iso, 0.0, 0.0, 0.0]
            names = [f"mega::frame_kernel<RGGB> (whole chain of {args.frames} frames in one launch: unpack + demosaic + statistics + "
                     "Reinhard + final map, grid barriers inside)", "-", "-", "-"]
            dom, dom_bytes = 0, ALG_BYTES * args.frames
        else:
            passes = time_passes(frames[0], bp.outputs[0], bp.ws.data_ptr(), device)
            names = PASS_NAMES
            # the dominant kernel: the pass that carries the frame's algorithmic bytes - it reads the packed frame and
            # writes the output (passes 0-2 only re-read the 18.9 MB packed frame and leave a few hundred bytes)
            dom, dom_bytes = 3, PASS_BYTES[3]
        dom_us = live[dom]
        achieved = dom_bytes / (dom_us * 1e-6) / 1e9
        # HBM-side bytes of the dominant kernel: NOT measured by this run - cited from the committed PMC profile
        # (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, scripts/profile_round.sh)
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                doc = json.load(open(tpath))
                k = doc["kernels"]["whole_frame" if whole else f"pass{dom}"]
                # per launch, like `achieved`: the profile's launch held frames_per_launch frames
                traffic = int(k["hbm_bytes_per_frame"] * args.frames) if whole and "hbm_bytes_per_frame" in k else k["hbm_bytes"]
                traffic_src = "profile-cited, not measured in this run: profiles/traffic_latest.json (" + doc.get("tag", "?") + ")"
            except Exception:
                traffic = None
        line = {
            "metric": METRIC,
            "value": round(value, 1), "unit": "MP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            **run_identity(world, backend),
            "us_per_frame_by_rank": per_rank,
            "config": {"workload": "config 2: 4096x3072 packed-12 RGGB -> demosaic -> Reinhard tonemap (stateless, "
                                   "gamma 1) -> f16 RGB", "frames_per_rank_per_step": args.frames,
                       "frames_per_step_all_ranks": frames_total,
                       "scaling_mode": ("weak: every rank processes --frames frames per step" if args.scaling == "weak" else
                                        "strong: --frames frames per step in total, split evenly over the ranks (BASELINE config 4: "
                                        "64 frames over 8 GPUs = 8 per launch)"),
                       "distinct_frames_per_rank": n_distinct,
                       "frame_seeds": f"1234 + (rank * frames_per_rank + i) % 64, i < {n_distinct} (SURVEY 8(d))"
                                      + ("" if n_distinct == args.frames else f", cycled over the {args.frames} buffers"),
                       "streams_per_rank": 1 if whole else args.streams, "work_dtype": "f16",
                       "chain": ("whole-frame kernel: one persistent launch per step (csrc/isp_mega.h), the resident grid walks "
                                 "through the step's frames" if whole else
                                 "multi-pass streaming chain (csrc/isp_stream.h), one frame per stream in flight"),
                       "launch": ("launch by launch, HIP events around every launch" if (whole or not use_graph) else
                                  f"HIP graph replay of the step (captured inside the library); every {args.eager_every}th "
                                  "step launch by launch for the per-pass events"),
                       "sharding": f"frames x{world}, no collective"},
            "us_per_frame": round(elapsed / (args.frames * args.steps) * 1e6, 2),
            "timed_region_s": round(elapsed, 3),
            "pipeline_frac_of_hbm_roofline": round(ALG_BYTES * frames_total * args.steps / elapsed / 1e9
                                                   / (HBM_PEAK_GBS * world), 4),
            "roofline": {"bound": "hbm", "kernel": names[dom],
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_us": round(dom_us, 2),
                         "launches_timed": int(live_n.value),
                         "frames_per_launch": args.frames if whole else 1,
                         "avg_us_per_frame": round(dom_us / (args.frames if whole else 1), 2),
                         "single_frame_launch_us": round(passes[dom], 2),
                         "single_frame_launch_frac": round(ALG_BYTES / (passes[dom] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if passes[dom] > 0 else None,
                         "note": "every pass of this chain is bound by instruction throughput (fetch + VALU issue), "
                                 "not by bandwidth: DESIGN.md 5"},
            "whole_frame_faults": faults,
            "kernels_us_live": ({"frame_kernel": round(live[0], 2)} if whole else {f"pass{k}": round(live[k], 2) for k in range(4)}),
            "kernels_us_isolated": ({"frame_kernel": round(passes[0], 2)} if whole else
                                    {f"pass{k}": round(passes[k], 2) for k in range(4)}),
        }
        if world == 1 and not args.no_other_workloads:
            del bp
            line["other_workloads"] = other_workloads(frames, host, device, args.frames)
            # the same kernel on frames whose bounds are not (0, 1) - what a sensor with a black level delivers: no
            # data-dependent shortcut (phase B and its barrier run)
            gp = line["other_workloads"].get("config2_whole_frame_kernel_bounds_not_unit")
            if gp and whole:
                line["roofline"]["general_path_frac"] = gp["frac_of_hbm_roofline"]
                line["roofline"]["general_path_us_per_frame"] = gp["us_per_frame"]
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
