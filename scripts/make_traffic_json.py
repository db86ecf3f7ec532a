"""profiles/<tag>_traffic.json (+ traffic_latest.json) from the FETCH_SIZE / WRITE_SIZE passes that
scripts/profile_round.sh leaves in gpurun_out/profile_<tag>/ .   usage: make_traffic_json.py r02"""
import csv, glob, json, subprocess, sys
tag = sys.argv[1]
d = f"gpurun_out/profile_{tag}"
# kernel name fragments: stream_kernel<half, 0, 0, EPI> of the multi-pass chain, frame_kernel<0, 0> = whole-frame kernel
KERNELS = [("pass0", "stream_kernelIDF16_Li0ELi0ELi1E"), ("pass1", "stream_kernelIDF16_Li0ELi0ELi2E"),
           ("pass2", "stream_kernelIDF16_Li0ELi0ELi3E"), ("pass3", "stream_kernelIDF16_Li0ELi0ELi4E"),
           ("whole_frame", "mega::frame_kernel<0, 0, false>|frame_kernelILi0ELi0ELb0E")]   # rocprofv3 demangles some names
FRAMES_PER_LAUNCH = {"whole_frame": int(sys.argv[2]) if len(sys.argv) > 2 else 64}   # scripts/prof_batch.py: 64 frames per launch
FETCH_FACTOR = 2   # scratch/fetch_calib.hip: FETCH_SIZE reports half the bytes for 4-, 12- and 16-byte-per-lane streams
def mean(sub, counter, kern):
    vals = []
    for p in glob.glob(f"{d}/{sub}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(p)):
            if row["Counter_Name"] == counter and any(k in row["Kernel_Name"] for k in kern.split("|")) and int(row["Grid_Size"]) > 100000:
                vals.append(float(row["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (0.0, 0)
out = {}
for name, kern in KERNELS:
    f, n = mean("pmc_fetch", "FETCH_SIZE", kern)
    w, _ = mean("pmc_write", "WRITE_SIZE", kern)
    if n == 0:
        continue
    out[name] = {"kernel": kern, "dispatches": n, "FETCH_SIZE_KB_raw": round(f, 1), "WRITE_SIZE_KB": round(w, 1),
                 "fetch_correction": FETCH_FACTOR,
                 "read_bytes": int(f * 1024 * FETCH_FACTOR), "write_bytes": int(w * 1024),
                 "hbm_bytes": int(f * 1024 * FETCH_FACTOR + w * 1024)}
    if name in FRAMES_PER_LAUNCH:
        n_fr = FRAMES_PER_LAUNCH[name]
        out[name]["frames_per_launch"] = n_fr
        out[name]["hbm_bytes_per_frame"] = out[name]["hbm_bytes"] // n_fr
        out[name]["read_bytes_per_frame"] = out[name]["read_bytes"] // n_fr
        out[name]["write_bytes_per_frame"] = out[name]["write_bytes"] // n_fr
try:
    commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
except Exception:
    commit = "?"
chain = [k for k in ("pass0", "pass1", "pass2", "pass3") if k in out]
doc = {"tag": f"{tag}, measured at commit {commit}",
       "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on scripts/prof_batch.py (64 frames per launch), MI355X",
       "note": "FETCH_SIZE on gfx950 reports half the bytes of wave-contiguous streaming reads; calibrated with "
               "scratch/fetch_calib.hip for 4-, 12- and 16-byte-per-lane loads (factor 2.000 each): doubled here. "
               "WRITE_SIZE is exact for 16-B/lane stores. Infinity-Cache hits are counted by these counters.",
       "kernels": out,
       "frame_hbm_bytes": sum(out[k]["hbm_bytes"] for k in chain),
       "whole_frame_kernel_hbm_bytes_per_launch": out.get("whole_frame", {}).get("hbm_bytes"),
       "whole_frame_kernel_hbm_bytes_per_frame": out.get("whole_frame", {}).get("hbm_bytes_per_frame"),
       "frame_algorithmic_bytes": 94371840}
for path in (f"profiles/{tag}_traffic.json", "profiles/traffic_latest.json"):
    json.dump(doc, open(path, "w"), indent=1)
print(json.dumps({k: v["hbm_bytes"] for k, v in out.items()}), doc["frame_hbm_bytes"])
