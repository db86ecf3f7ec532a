#!/bin/bash
# scripts/collect_profiles.sh TAG: copies what scripts/profile_round4.sh left under gpurun_out/profile_TAG/ into the tracked
# profiles/TAG_* files (headers name the commit the library was built from = HEAD) and rebuilds profiles/traffic_latest.json.
set -e
R=$(cd "$(dirname "$0")/.." && pwd); cd $R
TAG=${1:-r04}; P=gpurun_out/profile_$TAG; C=$(git rev-parse --short HEAD)
python scripts/make_traffic_json.py $TAG 64
cp $P/bench.json profiles/${TAG}_bench.json
cp $P/kernel_stats.csv profiles/${TAG}_bench_kernel_stats.csv
{ echo "# scripts/profile_round.sh $TAG (scripts/pmc_summary.py): counters of mega::frame_kernel per LAUNCH of 64 frames (scripts/prof_batch.py 64 3; rocprofv3 --pmc, separate passes), MI355X, commit $C"
  grep -v amdgpu.ids $P/pmc_summary.txt; } > profiles/${TAG}_pmc_summary.txt
{ echo "# scripts/profile_round4.sh (scripts/isp_pass_times.py on a rocprofv3 --kernel-trace of scripts/time_isp.py), MI355X, commit $C: every kernel of the 6-camera step, PER WORKLOAD"
  echo "# (round 3's table split by grid alone: the batched Reinhard passes use one grid for both image sizes, so its 35 - 142 us 'spread' was the two workloads in one line)"
  cat $P/isp_kernel_stats.txt; grep -v amdgpu $P/time_isp.txt; } > profiles/${TAG}_isp_kernel_stats.txt
{ echo "# scripts/wf_batch_stamps.py 6 on the stamps build (-DMI_STREAM_STAMPS -DMI_ISP_MEASURE -DMI_STAMP_REALTIME), MI355X, commit $C"
  grep -v amdgpu $P/whole_frame_timeline.txt; } > profiles/${TAG}_whole_frame_timeline.txt
{ echo "# scripts/wf_skew.py on the stamps build, MI355X, commit $C"
  grep -v amdgpu $P/whole_frame_skew.txt; } > profiles/${TAG}_whole_frame_skew.txt
{ echo "# scripts/pmc_mix.sh profile_${TAG}_mix 'prof_batch.py 64 3': dynamic instruction mix of mega::frame_kernel per LAUNCH of 64 frames (divide by 64 for a frame), the round's final kernel, MI355X, commit $C"
  sed -n '/mega::frame_kernel/,$p' $P/whole_frame_mix_batch64.txt; } > profiles/${TAG}_whole_frame_mix.txt
if [ -f $P/camera_group_kernel_stats.txt ]; then
{ echo "# scripts/profile_round4.sh: rocprofv3 --kernel-trace --stats of scripts/prof_cam.py (ISP.process_packed12, 6 cameras at 4096 x 3072, gamma 0.6, images not kept: 30 steps), then scripts/time_cam.py; MI355X, commit $C"
  grep -v amdgpu $P/camera_group_kernel_stats.txt; } > profiles/${TAG}_camera_group_kernel_stats.txt
fi
git status --short profiles | head -20
