#!/bin/bash
# Collects the round's evidence on the GPU box into gpurun_out/profile_<tag>/ :
#   bench.json            the default bench line
#   kernel_stats.csv      rocprofv3 --kernel-trace --stats of the same bench command
#   pmc_fetch / pmc_write FETCH_SIZE / WRITE_SIZE per dispatch of scripts/prof_batch.py: 64 frames per launch (separate passes)
#   pmc_valu / pmc_wait / pmc_lds   issue / wait / LDS counters of the same driver (separate passes)
# usage: scripts/profile_round.sh r02
set -o pipefail
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profile_$TAG
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err || exit 1
# the same command under the profiler, without the secondary workloads and the single-frame launches: every launch of
# mega::frame_kernel the trace holds is then a headline launch (64 frames), its average comparable to roofline.avg_launch_us
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline --no-other-workloads --no-isolated > $OUT/trace.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $R/scripts/prof_batch.py 64 3 > $OUT/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $R/scripts/prof_batch.py 64 3 > $OUT/pmc_write.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc_valu -- python3 $R/scripts/prof_batch.py 64 3 > $OUT/pmc_valu.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_wait -- python3 $R/scripts/prof_batch.py 64 3 > $OUT/pmc_wait.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU --output-format csv -d $OUT/pmc_lds -- python3 $R/scripts/prof_batch.py 64 3 > $OUT/pmc_lds.log 2>&1 || exit 1
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
python3 $R/scripts/pmc_summary.py $OUT > $OUT/pmc_summary.txt
tail -1 $OUT/bench.json | cut -c1-400
