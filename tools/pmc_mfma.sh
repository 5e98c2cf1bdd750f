#!/bin/bash
# matrix-core utilisation counters (own run, no tracing): tools/pmc_mfma.sh <script.py> <name> [kernel filter]
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${2:-pmc_mfma}
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 90 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_sq -- python3 $R/$1 > /dev/null 2> $O/pmc_sq.err
python3 $R/tools/pmc_counters.py $O/pmc_sq ${3:-} > $O/pmc_mfma.json
rm -rf $O/pmc_sq
cat $O/pmc_mfma.json
