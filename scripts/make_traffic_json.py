"""profiles/<tag>_traffic.json (+ traffic_latest.json) from the FETCH_SIZE / WRITE_SIZE passes that
scripts/profile_round.sh leaves in gpurun_out/profile_<tag>/ .   usage: make_traffic_json.py r01"""
import csv, glob, json, sys
tag = sys.argv[1]
d = f"gpurun_out/profile_{tag}"
KERNELS = [("pass0", "tile_kernelIDF16_Li0ELi0ELi5E", 1), ("pass1", "rgb_pass_kernelIDF16_hLi1E", 2),
           ("pass2", "rgb_pass_kernelIDF16_hLi2E", 2), ("pass3", "rgb_pass_kernelIDF16_DF16_Li3E", 2)]
def mean(sub, counter, kern):
    vals = []
    for p in glob.glob(f"{d}/{sub}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(p)):
            if row["Counter_Name"] == counter and kern in row["Kernel_Name"]:
                vals.append(float(row["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (0.0, 0)
out = {}
for name, kern, fetch_factor in KERNELS:
    f, n = mean("pmc_fetch", "FETCH_SIZE", kern)
    w, _ = mean("pmc_write", "WRITE_SIZE", kern)
    out[name] = {"kernel": kern, "dispatches": n, "FETCH_SIZE_KB_raw": round(f, 1), "WRITE_SIZE_KB": round(w, 1),
                 "fetch_correction": fetch_factor,
                 "read_bytes": int(f * 1024 * fetch_factor), "write_bytes": int(w * 1024),
                 "hbm_bytes": int(f * 1024 * fetch_factor + w * 1024)}
doc = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on scripts/prof_single.py 3, MI355X",
       "note": "FETCH_SIZE on gfx950 reports half the bytes of 16-B/lane streaming reads (MI355X_MICROARCH.md): the "
               "elementwise passes (16-B/lane loads) are doubled; pass0 reads the packed frame with 12-B/lane loads "
               "(global_load_dwordx3), an uncalibrated width, reported raw (20.0 MB for an 18.87 MB frame + tile halos; "
               "doubling would give an upper bound of 40 MB). WRITE_SIZE is exact for 16-B/lane stores. Infinity-Cache "
               "hits are counted by these counters.",
       "kernels": out,
       "pass0_hbm_bytes_per_launch": out["pass0"]["hbm_bytes"],
       "frame_hbm_bytes": sum(v["hbm_bytes"] for v in out.values()),
       "frame_algorithmic_bytes": 94371840}
for path in (f"profiles/{tag}_traffic.json", "profiles/traffic_latest.json"):
    json.dump(doc, open(path, "w"), indent=1)
print(json.dumps({k: v["hbm_bytes"] for k, v in out.items()}), doc["frame_hbm_bytes"])
