#!/bin/bash
# scripts/pmc_clock.sh LABEL [LIB]: clock, issue and instruction-cache counters of the whole-frame kernel (64-frame launches of
# scripts/prof_batch.py) for one build of the library -> gpurun_out/pmc_clock_LABEL.txt.  Two counter passes; the dispatch
# records carry start / end timestamps, so cycles / duration is the clock the kernel actually ran at.
set -o pipefail
LABEL=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
[ -n "$2" ] && export MI_ISP_LIB=$2
OUT=$R/gpurun_out/pmc_clock_$LABEL
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/p1 -- python3 $R/scripts/prof_batch.py 64 3 > $OUT/p1.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY --output-format csv -d $OUT/p2 -- python3 $R/scripts/prof_batch.py 64 3 > $OUT/p2.log 2>&1 || exit 1
python3 - $OUT > $R/gpurun_out/pmc_clock_$LABEL.txt <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(list); dur = []
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        if "frame_kernel" not in row["Kernel_Name"]: continue
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
        dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
d = sorted(dur)[len(dur) // 2]
print(f"launch duration (median, us): {d:.1f} = {d / 64:.2f} per frame")
for c in sorted(acc):
    v = sorted(acc[c])[len(acc[c]) // 2]
    print(f"  {c:24s} {v:16.0f}   per us {v / d:12.1f}")
PY
cat $R/gpurun_out/pmc_clock_$LABEL.txt
