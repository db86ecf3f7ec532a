"""From a rocprofv3 --kernel-trace of scripts/time_isp.py: average duration of every kernel of the 6-camera step, split by
grid (full resolution first, then resize_width=1920).
    python scripts/isp_pass_times.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"]
    short = ("P1 rgb_pass<5>" if "Li5EEE" in name else "P2 rgb_pass<6>" if "Li6EEE" in name else "metering<0>" if "metering_kernel" in name and "Li0EE" in name
             else "metering<1>" if "metering_kernel" in name else "metering (one launch)" if "metering_fused" in name else "load stream_kernel<S_STORE>" if "stream_kernel" in name else
             "load resize_kernel" if "resize_kernel" in name else "finalize" if "finalize_kernel" in name else None)
    if short is None: continue
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    acc[(short, r["Grid_Size_X"], r["Grid_Size_Y"])].append(dur)
for k in sorted(acc):
    v = acc[k]
    print(f"{k[0]:30s} grid {k[1]:>9s} x {k[2]:>2s}: n {len(v):4d}  avg {sum(v)/len(v):8.1f} us  min {min(v):8.1f}  max {max(v):8.1f}")
