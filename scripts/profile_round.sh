#!/bin/bash
# Collects the round's evidence on the GPU box into gpurun_out/profile_<tag>/ :
#   bench.json            the default bench line
#   kernel_stats.csv      rocprofv3 --kernel-trace --stats of the same bench command
#   pmc_fetch / pmc_write FETCH_SIZE / WRITE_SIZE per dispatch of the single-frame driver (separate passes)
# usage: scripts/profile_round.sh r01
set -o pipefail
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profile_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline > $OUT/trace.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/scripts/prof_single.py 3 > $OUT/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/scripts/prof_single.py 3 > $OUT/pmc_write.log 2>&1 || exit 1
find $OUT -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
tail -1 $OUT/bench.json
