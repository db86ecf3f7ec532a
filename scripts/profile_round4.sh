#!/bin/bash
# Round 4's evidence in one call on the GPU box (gpurun_out/profile_r04*/): scripts/profile_round.sh r04 (bench line, kernel
# stats of the same command, FETCH / WRITE / VALU / wait / LDS counters of 64-frame launches), the instruction mix of the same
# launches, the ISP path's kernels per workload, and the whole-frame kernel's timeline and skew (stamps build).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r04}
$R/scripts/profile_round.sh $TAG || exit 1
OUT=$R/gpurun_out/profile_$TAG
$R/scripts/pmc_mix.sh profile_${TAG}_mix "prof_batch.py 64 3" > $OUT/whole_frame_mix_batch64.txt 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/isp_trace -- python3 $R/scripts/time_isp.py > $OUT/isp_trace.log 2>&1 || exit 1
python3 $R/scripts/isp_pass_times.py $(find $OUT/isp_trace -name "*kernel_trace.csv" | head -1) > $OUT/isp_kernel_stats.txt || exit 1
cd $R
if [ -f $R/taichi_image_amd/lib/libmi355_isp_stamps.so ]; then
  MI_ISP_LIB=$R/taichi_image_amd/lib/libmi355_isp_stamps.so timeout -k 10 200 python3 $R/scripts/wf_batch_stamps.py 6 > $OUT/whole_frame_timeline.txt 2>&1 || exit 1
  MI_ISP_LIB=$R/taichi_image_amd/lib/libmi355_isp_stamps.so timeout -k 10 200 python3 $R/scripts/wf_skew.py > $OUT/whole_frame_skew.txt 2>&1 || true
fi
python3 $R/scripts/time_isp.py > $OUT/time_isp.txt 2>&1
# the camera-group path (ISP.process_packed12): its three kernels per 6-camera step, and the step's time beside the two-call sequence
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/cam_trace -- python3 $R/scripts/prof_cam.py 0 > $OUT/cam_trace.log 2>&1 || exit 1
cd $R
find $OUT/cam_trace -name "*kernel_stats.csv" -exec cat {} + < /dev/null | cut -c1-160 | head -6 > $OUT/camera_group_kernel_stats.txt
python3 $R/scripts/time_cam.py >> $OUT/camera_group_kernel_stats.txt 2>&1
cat $OUT/isp_kernel_stats.txt
