#!/bin/bash
# PMC counters of one data pass of the config-2 chain (scripts/prof_pass.py), two counter groups in separate runs.
# usage: scripts/pmc_pass.sh <pass 0..3> <tag>
set -o pipefail
P=${1:-0}; TAG=${2:-pmc}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $OUT/a -- python3 $R/scripts/prof_pass.py $P 5 > $OUT/a.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/b -- python3 $R/scripts/prof_pass.py $P 5 > $OUT/b.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/c -- python3 $R/scripts/prof_pass.py $P 5 > $OUT/c.log 2>&1 || exit 1
python3 $R/scripts/pmc_summary.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
