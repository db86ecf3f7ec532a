#!/bin/bash
# scripts/build_variant.sh NAME "-DFLAG ..." [all] : a copy of the library compiled under extra flags
# (taichi_image_amd/lib/variants/libv_NAME.so).  Default: only the whole-frame kernel's objects are rebuilt (the others
# are taken from the main build); "all": everything.  Measurement only.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; FLAGS=$2; WHAT=${3:-mega}
mkdir -p $R/taichi_image_amd/lib/variants
rm -rf $R/build/v_$NAME
if [ "$WHAT" = "all" ]; then mkdir -p $R/build/v_$NAME; else cp -rp $R/build/csrc $R/build/v_$NAME && rm -f $R/build/v_$NAME/isp_mega_p*.o; fi
make -s -C $R/taichi_image_amd/csrc -j8 EXTRA="$FLAGS" OBJDIR=../../build/v_$NAME OUT=../lib/variants/libv_$NAME.so 2>&1 | grep -E "error" -A3 || true
ls -la $R/taichi_image_amd/lib/variants/libv_$NAME.so
