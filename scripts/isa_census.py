"""Static census of the whole-frame kernel's ISA, phase by phase.

Compiles csrc/isp_mega_p0.hip to gfx950 assembly with -DMI_MEGA_CENSUS (isp_mega.h: the run-time choices of the headline
configuration become constants, every phase boundary leaves a "; MI_MARK n" comment between two scheduling barriers) and
counts the instructions of frame_kernel<0, 0, false> between the marks, by issue class.  Straight-line code: one count =
one execution per wave (the poll loops of the barrier folds are counted once).

    python scripts/isa_census.py [extra -D flags ...] > profiles/r03_whole_frame_census.txt
"""
import collections, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "taichi_image_amd", "csrc")
PHASES = {0: "prologue (LUT build, first loads, 4 rows decoded)", 10: "phase A: 12 rows (decode, demosaic, bounds, f16, statistics, residency)",
          1: "post 0 (wave reductions, record)", 2: "barrier 0 (watch, fold, finalize)", 3: "phase B (skipped: bounds (0,1)) + scalars",
          4: "phase C: Reinhard of 12 rows, bounds", 5: "post 2", 6: "barrier 2", 7: "phase D: Reinhard of 5 LDS rows, final map of 12, stores",
          8: "epilogue"}

def classify(op):
    if op.startswith("v_"):
        if re.match(r"v_(log|exp|rcp|rsq|sqrt|sin|cos)_", op): return "valu trans"
        if re.match(r"v_(fma|fmac|fmamk|fmaak|mad|mac)_(f32|legacy)", op) or op.startswith("v_fma_mix"): return "valu fma" if not op.startswith("v_fma_mix") else "valu fma_mix"
        if re.match(r"v_pk_(fma|mul|add)_f32", op): return "valu pk_f32"
        if re.match(r"v_mul_(f32|legacy)", op): return "valu mul"
        if re.match(r"v_(add|sub|subrev)_f32", op): return "valu add"
        if re.match(r"v_cvt_", op): return "valu cvt"
        if re.match(r"v_(min|max|med)3?_|v_pk_(min|max)|v_(minimum|maximum)", op): return "valu min/max/med3"
        if re.match(r"v_cndmask", op): return "valu cndmask"
        if re.match(r"v_cmp", op): return "valu cmp"
        if re.match(r"v_mov_b32|v_accvgpr|v_mov_b64", op): return "valu mov"
        if re.match(r"v_(readlane|readfirstlane|writelane)", op): return "valu lane"
        if re.match(r"v_pk_", op): return "valu pk other"
        if re.match(r"v_.*_f64|v_cvt_f64", op): return "valu f64"
        return "valu int/bit"
    if op.startswith("ds_"): return "lds"
    if op.startswith("buffer_") or op.startswith("global_") or op.startswith("flat_") or op.startswith("scratch_"): return "vmem"
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith("s_load") or op.startswith("s_buffer_load"): return "smem"
    if re.match(r"s_(cbranch|branch)", op): return "branch"
    if op.startswith("s_nop") or op.startswith("s_sleep"): return "s_nop/sleep"
    if op.startswith("s_"): return "salu"
    return "other"

def main():
    extra = sys.argv[1:]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++20", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-math-errno",
               "-fno-slp-vectorize", "-S", "--cuda-device-only", "-DMI_MEGA_CENSUS", *extra, os.path.join(CSRC, "isp_mega_p0.hip"), "-o", out]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        text = open(out).read()
    name = "_ZN4mega12frame_kernelILi0ELi0ELb0EEEvNS_6MBatchE"
    body = text[text.index(name + ":"):]
    body = body[:body.index(".Lfunc_end0:")]
    lines = body.splitlines()
    # The compiler lays cold / conditional blocks out behind the first s_endpgm: a block belongs to the phase of the branch
    # that first reaches it.  Pass 1 learns label -> phase from the branch sites, pass 2 counts.
    label_phase = {}
    for _ in range(4):
        cur, after_end = 0, False
        for line in lines:
            m = re.search(r"; MI_MARK (\d+)", line)
            if m: cur = int(m.group(1)); continue
            t = line.strip()
            lm = re.match(r"^(\.LBB0_\d+):", t)
            if lm:
                if after_end and lm.group(1) in label_phase: cur = label_phase[lm.group(1)]
                continue
            if t.startswith("s_endpgm"): after_end = True; continue
            bm = re.match(r"^s_c?branch\S*\s+(\.LBB0_\d+)", t)
            if bm and bm.group(1) not in label_phase: label_phase[bm.group(1)] = cur
    cur, after_end = 0, False
    counts = collections.OrderedDict((k, collections.Counter()) for k in PHASES)
    dpp = collections.Counter(); sdwa = collections.Counter(); ops = collections.defaultdict(collections.Counter)
    for line in lines:
        m = re.search(r"; MI_MARK (\d+)", line)
        if m:
            cur = int(m.group(1)); continue
        t = line.strip()
        lm = re.match(r"^(\.LBB0_\d+):", t)
        if lm:
            if after_end and lm.group(1) in label_phase: cur = label_phase[lm.group(1)]
            continue
        if t.startswith("s_endpgm"): after_end = True; continue
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"): continue
        op = t.split()[0]
        if not re.match(r"^[a-z]", op): continue
        counts[cur][classify(op)] += 1
        ops[cur][op] += 1
        if "row_" in t or "quad_perm" in t or "wave_sh" in t or "dpp" in op: dpp[cur] += 1
        if "sdwa" in op or "_sdwa" in t or "src0_sel" in t: sdwa[cur] += 1
    classes = sorted({c for k in counts for c in counts[k]}, key=lambda c: (not c.startswith("valu"), c))
    px = 12 * 512 / 64     # pixels per lane of a wave: 12 rows x 8
    print("# scripts/isa_census.py: static instruction census of mega::frame_kernel<0,0,false>, headline configuration")
    print("# (interior band, bounds (0,1), color_adapt 0, f16 out); counts per WAVE (2048 waves per 4K frame, 96 pixels per lane)")
    if extra: print("# extra flags:", " ".join(extra))
    tot_all = collections.Counter()
    for k, title in PHASES.items():
        c = counts[k]
        valu = sum(v for cl, v in c.items() if cl.startswith("valu"))
        print(f"\n[{k}] {title}: {sum(c.values())} instructions, {valu} VALU = {valu / px:.1f} per pixel  (DPP {dpp[k]}, SDWA {sdwa[k]})")
        for cl in classes:
            if c[cl]: print(f"    {cl:22s} {c[cl]:6d}  {c[cl] / px:7.2f} /px")
        top = ", ".join(f"{o} {n}" for o, n in ops[k].most_common(14))
        print(f"    top: {top}")
        tot_all.update(c)
    valu = sum(v for cl, v in tot_all.items() if cl.startswith("valu"))
    print(f"\n[all] {sum(tot_all.values())} instructions, {valu} VALU per wave = {valu / px:.1f} per pixel; x 2048 waves = {valu * 2048 / 1e6:.2f} M wave-instructions")
    for cl in classes:
        print(f"    {cl:22s} {tot_all[cl]:6d}  {tot_all[cl] / px:7.2f} /px")

if __name__ == "__main__":
    main()
