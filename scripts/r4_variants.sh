#!/bin/bash
# scripts/r4_variants.sh OUTDIR [n_frames]: scripts/wf_time.py on the shipped library and on every variant in taichi_image_amd/lib/variants
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1; N=${2:-64}
mkdir -p $OUT
python3 $R/scripts/wf_time.py $N default >> $OUT/wf_time.txt 2>>$OUT/wf_time.err || exit 1
for so in $R/taichi_image_amd/lib/variants/libv_*.so; do
  case $so in *stamps*) continue;; esac
  MI_ISP_LIB=$so timeout -k 10 120 python3 $R/scripts/wf_time.py $N >> $OUT/wf_time.txt 2>>$OUT/wf_time.err || exit 1
done
cat $OUT/wf_time.txt
