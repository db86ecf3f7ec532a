#!/bin/bash
# VALU / LDS counters and kernel time of rstrm::resize_kernel (scripts/prof_resize.py: Camera16(resize_width=1920).load_packed12
# of a 4K frame), round 3: compare with profiles/r02_resize_pmc.txt ("after" column = the round-2 kernel).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_resize3
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc -- python3 $R/scripts/prof_resize.py 5 > $OUT/pmc.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $OUT/pmc2 -- python3 $R/scripts/prof_resize.py 5 > $OUT/pmc2.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/scripts/prof_resize.py 20 > $OUT/trace.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for path in glob.glob("$OUT/pmc*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        if "resize" in row["Kernel_Name"]:
            acc[(row["Kernel_Name"][:60], row["Counter_Name"])].append(float(row["Counter_Value"]))
for path in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        if "resize" in row["Name"]:
            print(row["Name"][:60], "avg ns", row["AverageNs"], "calls", row["Calls"])
for (k, c), v in sorted(acc.items()):
    print(k, c, round(sum(v) / len(v), 1))
PY
