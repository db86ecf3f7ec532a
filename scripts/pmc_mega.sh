#!/bin/bash
# PMC counters of the whole-frame kernel (scripts/mega_check.py), instruction cache + issue counters in separate runs.
set -o pipefail
TAG=${1:-pmc_mega}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $OUT/a -- python3 $R/scripts/mega_check.py > $OUT/a.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_IFETCH --output-format csv -d $OUT/b -- python3 $R/scripts/mega_check.py > $OUT/b.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/c -- python3 $R/scripts/mega_check.py > $OUT/c.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(path)):
        if "frame_kernel" in row["Kernel_Name"] and int(row["Grid_Size"]) > 100000:
            acc[row["Kernel_Name"][:50]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in acc:
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]; print(f"   {c:30s} {sum(v)/len(v):16.1f} (n={len(v)})")
PY
