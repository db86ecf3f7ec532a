"""Summarise rocprofv3 --pmc counter_collection csv files: per kernel name, mean of each counter."""
import csv
import glob
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        name = row["Kernel_Name"]
        if "frame_kernel" not in name and "stream_kernel" not in name and "tile_kernel" not in name and "finalize" not in name and "rgb_pass" not in name and "metering" not in name:
            continue
        short = name[:75]
        acc[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"    {c:28s} {sum(v) / len(v):16.1f}  (n={len(v)})")
