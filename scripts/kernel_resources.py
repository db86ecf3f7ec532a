"""Per-kernel resource usage of libmi355_isp.so, read from the code objects it carries (no compilation):
    python scripts/kernel_resources.py [lib.so]  ->  name, VGPRs, SGPR spills, VGPR spills, scratch bytes, LDS bytes
Used by tests/test_abi.py: no kernel of the library may touch scratch."""
import os, re, subprocess, sys, tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def kernel_resources(lib_path):
    res = {}
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", lib_path, os.path.join(d, "copy.so")], check=True)
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        for i, s0 in enumerate(starts):
            part = os.path.join(d, f"b{i}.bin")
            with open(part, "wb") as f:
                f.write(blob[s0:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            co = os.path.join(d, f"co{i}.elf")
            r = subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                                f"--input={part}", f"--output={co}"], capture_output=True)
            if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            # kernel descriptors (<name>.kd, 64 bytes in .rodata): bit 1 of kernel_code_properties (byte 56) = the kernel takes
            # the address of its dispatch packet - i.e. it reads the packet, which lives in HOST memory (13 - 27 us away:
            # a local array indexed by the thread id did that to the one-launch metering kernel in round 3)
            dispatch_ptr = {}
            syms = subprocess.run([f"{LLVM}/llvm-readelf", "-sW", co], capture_output=True, text=True).stdout
            secs = subprocess.run([f"{LLVM}/llvm-readelf", "-SW", co], capture_output=True, text=True).stdout
            m = re.search(r"\.rodata\s+PROGBITS\s+([0-9a-f]+)\s+([0-9a-f]+)\s+([0-9a-f]+)", secs)
            if m:
                ro_addr, ro_off = int(m.group(1), 16), int(m.group(2), 16)
                image = open(co, "rb").read()
                for line in syms.splitlines():
                    f = line.split()
                    if len(f) >= 8 and f[-1].endswith(".kd"):
                        off = ro_off + int(f[1], 16) - ro_addr
                        props = int.from_bytes(image[off + 56:off + 58], "little")
                        dispatch_ptr[f[-1][:-3]] = bool(props & 2)
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
            for blk in notes.split("- .agpr_count:")[1:]:
                name = re.search(r"\.name:\s+(\S+)", blk)
                if not name:
                    continue
                g = lambda key: int(re.search(rf"\.{key}:\s+(\d+)", blk).group(1))
                res[name.group(1)] = {"vgprs": g("vgpr_count"), "sgpr_spills": g("sgpr_spill_count"), "vgpr_spills": g("vgpr_spill_count"),
                                      "scratch": g("private_segment_fixed_size"), "lds": g("group_segment_fixed_size"),
                                      "dispatch_ptr": dispatch_ptr.get(name.group(1), False)}
    return res


if __name__ == "__main__":
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "taichi_image_amd", "lib", "libmi355_isp.so")
    r = kernel_resources(lib)
    print(f"{len(r)} kernels in {lib}")
    for k, v in sorted(r.items(), key=lambda kv: -kv[1]["scratch"]):
        if v["scratch"] or v["vgpr_spills"] or v["dispatch_ptr"] or "-a" in sys.argv:
            print(f"  {k[:100]:100s} vgprs {v['vgprs']:3d} scratch {v['scratch']:5d} vgpr spills {v['vgpr_spills']:4d} lds {v['lds']} dispatch packet read: {v['dispatch_ptr']}")
