#!/bin/bash
# SQ counter pass (own run, no tracing) of one probe script: tools/pmc_one.sh <script.py> <name> [kernel-name filter]
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${2:-pmc1}
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 90 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_sq -- python3 $R/$1 > /dev/null 2> $O/pmc_sq.err
python3 $R/tools/pmc_counters.py $O/pmc_sq ${3:-} > $O/pmc_sq.json
rm -rf $O/pmc_sq
cat $O/pmc_sq.json
