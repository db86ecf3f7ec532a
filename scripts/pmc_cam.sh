#!/bin/bash
# FETCH_SIZE / WRITE_SIZE / VALU counters of the camera-group path's kernels (scripts/prof_cam.py KEEP: 30 steps of six 4K
# cameras), separate rocprofv3 --pmc passes -> gpurun_out/<dir>/summary.txt (per kernel, mean per launch).
# usage: scripts/pmc_cam.sh pmc_cam [keep_images 0|1]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-pmc_cam}; KEEP=${2:-0}
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/scripts/prof_cam.py $KEEP > $OUT/fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/scripts/prof_cam.py $KEEP > $OUT/write.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $OUT/valu -- python3 $R/scripts/prof_cam.py $KEEP > $OUT/valu.log 2>&1 || exit 1
python3 - $OUT <<'PY' > $OUT/summary.txt
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        n = row["Kernel_Name"]
        if "camera_kernel" in n or "sub_kernel" in n or "metering" in n:
            acc[n[:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"    {c:24s} {sum(v) / len(v):16.1f}  (n={len(v)})")
PY
cat $OUT/summary.txt
