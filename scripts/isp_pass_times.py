"""From a rocprofv3 --kernel-trace of scripts/time_isp.py: duration of every kernel of the 6-camera step, PER WORKLOAD - the
script runs the full-resolution group first, then resize_width=1920; the two batched Reinhard passes use the same grid for
both image sizes, so round 3's table (split by grid alone) averaged 4096 x 3072 and 1440 x 1920 launches into one line and
called the difference a spread.  A launch belongs to the workload of the last load kernel before it.
    python scripts/isp_pass_times.py <kernel_trace.csv>"""
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
acc = collections.defaultdict(list)
mode = "?"
for r in rows:
    name = r["Kernel_Name"]
    short = ("P1 rgb_pass<5>" if "Li5EEE" in name else "P2 rgb_pass<6>" if "Li6EEE" in name else "metering<0>" if "metering_kernel" in name and "Li0EE" in name
             else "metering<1>" if "metering_kernel" in name else "metering (one launch)" if "metering_fused" in name else "load stream_kernel<S_STORE>" if "stream_kernel" in name else
             "load resize_kernel" if "resize_kernel" in name else "finalize" if "finalize_kernel" in name else
             "tonemap (one launch)" if "reinhard_fused" in name else None)
    if short is None: continue
    if short.startswith("load"):
        mode = "6 x 4096x3072" if "stream_kernel" in name else "6 x 1440x1920 (resize_width=1920)"
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    acc[(mode, short, r["Grid_Size_X"], r["Grid_Size_Y"])].append(dur)
for k in sorted(acc):
    v = sorted(acc[k])
    med = v[len(v) // 2]
    print(f"{k[0]:34s} {k[1]:30s} grid {k[2]:>9s} x {k[3]:>2s}: n {len(v):4d}  avg {sum(v)/len(v):8.1f} us  median {med:8.1f}  min {v[0]:8.1f}  max {v[-1]:8.1f}  max/min {v[-1]/v[0]:.2f}")
