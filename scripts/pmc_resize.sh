#!/bin/bash
# LDS / VALU counters of the fused load + resize kernels, before (round-1 tile kernel) and after (streaming kernel).
# Needs a measure build for the "before" leg: make -C taichi_image_amd/csrc EXTRA=-DMI_ISP_MEASURE OBJDIR=../../build/csrc_measure OUT=../lib/libmi355_isp_measure.so
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_resize
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MI_ISP_LIB=$R/taichi_image_amd/lib/libmi355_isp_measure.so
for leg in after before; do
  if [ $leg = before ]; then export MI_ISP_NO_STREAM=1; else unset MI_ISP_NO_STREAM; fi
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/$leg -- python3 $R/scripts/prof_resize.py 5 > $OUT/$leg.log 2>&1 || exit 1
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${leg}_trace -- python3 $R/scripts/prof_resize.py 20 > $OUT/${leg}_trace.log 2>&1 || exit 1
done
python3 - <<PY
import csv, glob, collections
for leg in ("before", "after"):
    acc = collections.defaultdict(list)
    for path in glob.glob("$OUT/%s/**/*counter_collection.csv" % leg, recursive=True):
        for row in csv.DictReader(open(path)):
            if "resize" in row["Kernel_Name"]:
                acc[(row["Kernel_Name"][:60], row["Counter_Name"])].append(float(row["Counter_Value"]))
    for path in glob.glob("$OUT/%s_trace/**/*kernel_stats.csv" % leg, recursive=True):
        for row in csv.DictReader(open(path)):
            if "resize" in row["Name"]:
                print(leg, row["Name"][:60], "avg ns", row["AverageNs"], "calls", row["Calls"])
    for (k, c), v in sorted(acc.items()):
        print(leg, k, c, round(sum(v) / len(v), 1))
PY
