#!/bin/bash
# Order of the images in the two batched passes of ISP.tonemap_reinhard (mi_isp_reinhard_batch): f = list order, r = reversed;
# first letter pass 1, second pass 2.  6 cameras, 4096x3072 and 1920x1440 (scripts/time_isp.py), measure build.
R=${GRAFT_REPO_ROOT:-$(pwd)}
export MI_ISP_LIB=$R/taichi_image_amd/lib/libmi355_isp_measure.so
for o in ff rf fr rr; do echo "== MI_ISP_ORDER=$o"; MI_ISP_ORDER=$o python3 $R/scripts/time_isp.py 2>&1 | grep -v amdgpu.ids; done
