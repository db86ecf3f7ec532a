#!/bin/bash
# kernel durations of the 6-camera ISP step per workload (rocprofv3 --kernel-trace of scripts/time_isp.py -> scripts/isp_pass_times.py)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-isp_trace}
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/scripts/time_isp.py > $OUT/trace.log 2>&1 || exit 1
python3 $R/scripts/isp_pass_times.py $(find $OUT/trace -name "*kernel_trace.csv" | head -1) | tee $OUT/isp_kernel_stats.txt
grep -v amdgpu $OUT/trace.log
