"""The C-ABI library loads on a GPU-less host and exports every symbol include/mi_isp.h
declares; host-only entry points behave; argument validation fails loudly (no kernel runs)."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle import isp_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mi_isp.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mi_isp_\w+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    from taichi_image_amd import _native
    L = _native.lib()
    names = declared_symbols()
    assert len(names) >= 22
    for n in names:
        assert hasattr(L, n), f"{n} missing from libmi355_isp.so"
        assert n in _native.SIGNATURES, f"{n} has no ctypes signature"
    assert sorted(_native.SIGNATURES) == names


def test_host_only_entry_points():
    from taichi_image_amd import _native, bayer
    L = _native.lib()
    assert L.mi_isp_version() >= 1000
    assert np.array_equal(bayer.bayer_weights(), O.BAYER_KERNELS)
    assert L.mi_isp_workspace_bytes(3072, 4096) >= 64 * 4 + 8 * 3072 * 4
    assert L.mi_isp_workspace_bytes(0, 0) == 0


def test_argument_validation_reports_errors():
    from taichi_image_amd import _native
    L = _native.lib()
    assert L.mi_isp_decode12(None, None, 4, 1, 0, 0, None) != 0
    assert b"null" in L.mi_isp_last_error()
    buf = (ctypes.c_uint8 * 16)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert L.mi_isp_decode12(p, p, 3, 1, 0, 0, None) != 0           # odd pixel count
    assert b"even" in L.mi_isp_last_error()
    assert L.mi_isp_demosaic(p, p, 5, 8, 2, 2, 0, None, None) != 0   # odd height (bayer.py:206)
    assert b"even size" in L.mi_isp_last_error()
    assert L.mi_isp_demosaic(p, p, 4, 8, 9, 2, 0, None, None) != 0
    assert L.mi_isp_transform(p, p, 4, 8, 0, 7, None) != 0           # transverse on non-square
    with pytest.raises(RuntimeError, match="libmi355_isp"):
        _native.check(L.mi_isp_resize_bilinear(p, p, 4, 4, 2, 2, 0.0, 0.5, 2, 2, None))
    # the round-3 entry points validate before they touch a device
    P = ctypes.c_void_p
    one = (P * 1)(p)
    assert L.mi_isp_pipeline12_reinhard_whole_frame_batch(None, None, 1, 64, 512, 0, 0, None, 2, 1.0, 1.0, 1.0, 0.0, None, None) != 0
    assert b"null" in L.mi_isp_last_error()
    assert L.mi_isp_pipeline12_reinhard_whole_frame_batch(one, one, 0, 64, 512, 0, 0, None, 2, 1.0, 1.0, 1.0, 0.0, p, None) != 0
    assert b"at least one frame" in L.mi_isp_last_error()
    assert L.mi_isp_load_packed_batch(None, None, None, 1, 64, 512, 12, 0, 0, None, 2, 64, 512, 0.0, 8, None) != 0
    assert L.mi_isp_load_packed_batch(one, one, None, -1, 64, 512, 12, 0, 0, None, 2, 64, 512, 0.0, 8, None) != 0
    assert L.mi_isp_load_packed_batch(one, one, None, 0, 64, 512, 12, 0, 0, None, 2, 64, 512, 0.0, 8, None) == 0      # nothing to do
    assert L.mi_isp_load_packed_metered(p, p, 64, 512, 12, 0, 0, None, 2, 64, 512, 0.0, None, 8, None) != 0           # no subsample buffer
    nbad = ctypes.c_int(0)
    assert L.mi_isp_workspace_check(None, 1, 64, 512, None, ctypes.byref(nbad), None) != 0
    assert L.mi_isp_workspace_check(p, 1, 0, 0, None, ctypes.byref(nbad), None) != 0
    assert L.mi_isp_load_packed_metered_is_fused(3072, 4096, 12, 0, 2, 8) == 1
    assert L.mi_isp_load_packed_metered_is_fused(3072, 4096, 12, 0, 2, 4) == 0 and L.mi_isp_load_packed_metered_is_fused(70, 204, 12, 0, 2, 8) == 0
    assert L.mi_isp_load_packed_metered_is_fused(3072, 4096, 16, 0, 2, 8) == 0 and L.mi_isp_load_packed_metered_is_fused(3072, 4096, 12, 1, 2, 8) == 0
    assert L.mi_isp_whole_frame_set_poll_limit(0) == 0
    # the camera-group entry points (round 4) validate before they touch a device too
    f = ctypes.c_float
    assert L.mi_isp_camera_group_reinhard(None, None, None, 1, 64, 512, 0, None, None, None, f(0), f(1), f(1), f(1), f(0), None, None, None) != 0
    assert b"null" in L.mi_isp_last_error()
    assert L.mi_isp_camera_group_reinhard(one, None, one, 0, 64, 512, 0, None, p, p, f(0), f(1), f(1), f(1), f(0), p, p, None) != 0
    assert b"cameras per call" in L.mi_isp_last_error()
    assert L.mi_isp_camera_group_subsample(one, 65, 64, 512, 0, None, p, None) != 0
    assert L.mi_isp_camera_group_tonemap(one, None, one, 1, 64, 512, 0, None, p, f(0), f(1), f(1), f(0), p, None) != 0      # gamma 0
    assert b"gamma" in L.mi_isp_last_error()
    assert L.mi_isp_camera_group_scratch_bytes(6, 3072, 4096) == 6 * 384 * 512 * 3 * 2
    assert L.mi_isp_camera_group_scratch_bytes(1, 70, 204) == (9 * 26 * 3 * 2 + 255) // 256 * 256 and L.mi_isp_camera_group_scratch_bytes(0, 8, 8) == 0
    assert L.mi_isp_camera_group_fits(3072, 4096, 0, 3, 8) == 0 and L.mi_isp_camera_group_fits(3072, 4096, 0, 2, 4) == 0      # f32 / stride 4: no
    assert L.mi_isp_camera_group_set_poll_limit(0) == 0


def test_no_cpu_fallback():
    """Without a GPU the public ops raise instead of computing on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import taichi_image_amd as ti
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ti.packed.decode12(np.zeros(6, np.uint8))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ti.bayer.bayer_to_rgb(np.zeros((4, 4), np.float32))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "taichi_image_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".inc")):
                text = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in text and "from oracle" not in text, f
    # bench.py may touch the oracle only inside its cpu_baseline leg
    head = open(os.path.join(ROOT, "bench.py")).read().split("def cpu_baseline")[0]
    assert "import oracle" not in head and "from oracle" not in head


def test_dtype_tokens():
    import torch
    from taichi_image_amd import types
    assert types.as_dtype("f16") is types.f16 and types.as_dtype(np.float16) is types.f16
    assert types.as_dtype(torch.float16) is types.f16 and types.as_dtype(np.dtype("uint16")) is types.u16
    assert types.as_dtype("float32") is types.f32 and types.as_dtype(torch.uint8) is types.u8
    with pytest.raises(KeyError):
        types.as_dtype(np.float64)
    with pytest.raises(ValueError):
        types.ti_type([1, 2, 3])
    assert types.scale_factor[types.u16] == 65535


def test_native_calls_run_on_the_device_of_their_stream(monkeypatch):
    """a call whose stream belongs to a GPU other than the current one must run under that device
    (pointers, workspace and launch then agree); unit test of the guard with a fake entry point."""
    import torch
    from taichi_image_amd import _native
    entered = []

    class FakeGuard:
        def __init__(self, dev): self.dev = dev
        def __enter__(self): entered.append(self.dev)
        def __exit__(self, *a): entered.append("exit")

    monkeypatch.setattr(torch.cuda, "device", FakeGuard)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    calls = []
    fn = _native._Guarded(lambda *a: calls.append(a) or 0)
    s1 = _native._StreamArg(0); s1.device = torch.device("cuda", 1)
    s0 = _native._StreamArg(0); s0.device = torch.device("cuda", 0)
    assert fn(1, 2, s1) == 0 and entered == [torch.device("cuda", 1), "exit"]
    entered.clear()
    assert fn(1, 2, s0) == 0 and fn(3) == 0 and entered == []          # current device / no stream: no switch
    assert len(calls) == 3 and int(calls[0][2]) == 0


def test_whole_frame_kernel_uses_no_scratch():
    """csrc/isp_mega.h sits at the 256-VGPR limit (two blocks per CU is what keeps a 4K frame resident): a spill to
    scratch costs microseconds per frame (measured 3.3) and is decided by the register allocator PER INSTANTIATION, so
    the build is checked: the device assembly of the kernels of all four CFA patterns (x color_adapt == 0 / != 0) must
    report no spilled VGPR and no private segment."""
    import re, shutil, subprocess, tempfile
    from concurrent.futures import ThreadPoolExecutor
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def one(i):
        src = os.path.join(root, "taichi_image_amd", "csrc", f"isp_mega_p{i}.hip")
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "mega.s")
            subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++20", "-ffp-contract=off", "-fno-math-errno",
                            "-fno-slp-vectorize", "--cuda-device-only", "-S", "-I" + os.path.join(root, "include"), "-o", out, src],
                           check=True, capture_output=True)
            text = open(out).read()
        kernels = re.findall(r"\.name:\s+(\S*frame_kernel\S*)", text)
        spills = [int(x) for x in re.findall(r"\.vgpr_spill_count:\s+(\d+)", text)]
        scratch = [int(x) for x in re.findall(r"\.private_segment_fixed_size:\s+(\d+)", text)]
        return kernels, spills, scratch

    with ThreadPoolExecutor(4) as ex:
        results = list(ex.map(one, range(4)))
    for i, (kernels, spills, scratch) in enumerate(results):
        assert len(kernels) == 2 and len(spills) == 2 and len(scratch) == 2, (i, kernels, spills, scratch)
        assert spills == [0, 0] and scratch == [0, 0], (i, kernels, spills, scratch)


def test_no_kernel_of_the_library_touches_scratch():
    """Read from the code objects inside libmi355_isp.so (scripts/kernel_resources.py): no kernel spills registers and
    none has a private segment.  Besides spills this catches by-value argument structs that the compiler copies to scratch
    to index them - a computed index into the batched passes' pointer lists did that in round 3 and cost 2.4 x."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"):
        pytest.skip("llvm tools not available")
    spec = importlib.util.spec_from_file_location("kernel_resources", os.path.join(root, "scripts", "kernel_resources.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    from taichi_image_amd import _native
    res = mod.kernel_resources(_native.LIB_PATH)
    assert len(res) > 300, len(res)
    assert any("frame_kernel" in k for k in res) and any("rgb_pass_kernel" in k for k in res)
    bad = {k: v for k, v in res.items() if v["scratch"] or v["vgpr_spills"]}
    assert not bad, bad
    # ... and none reads its dispatch packet (host memory): the one-launch metering kernel did, for a local array indexed by
    # the thread id that the compiler moved to LDS slots addressed by the flat thread id - 20 of its 34 us
    reads = [k for k, v in res.items() if v["dispatch_ptr"]]
    assert not reads, reads
