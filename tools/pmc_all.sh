#!/bin/bash
# Counter passes of tools/pmc_probe.py (each --pmc pass is a run of its own, never combined with tracing):
# kernel trace, FETCH_SIZE, WRITE_SIZE -> pmc_hot_kernels.json; SQ counters of the FPS kernels -> pmc_fps.json.
# Results under gpurun_out/$1; copy what is to be judged into profiles/.
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-pmc}
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_probe -- python3 $R/tools/pmc_probe.py > $O/probe.json 2> $O/probe.err
timeout -s KILL 90 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/tools/pmc_probe.py > /dev/null 2> $O/pmc_fetch.err
timeout -s KILL 90 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/tools/pmc_probe.py > /dev/null 2> $O/pmc_write.err
python3 $R/tools/pmc_summary.py $O/trace_probe $O/pmc_fetch $O/pmc_write > $O/pmc_hot_kernels.json
cp $O/trace_probe/*/*_kernel_stats.csv $O/probe_kernel_stats.csv
timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_sq -- python3 $R/tools/pmc_probe.py > /dev/null 2> $O/pmc_sq.err
python3 $R/tools/pmc_counters.py $O/pmc_sq fps,fb_sample > $O/pmc_fps.json
python3 $R/tools/pmc_counters.py $O/pmc_sq > $O/pmc_sq_all.json
rm -rf $O/pmc_fetch $O/pmc_write $O/trace_probe $O/pmc_sq
cat $O/pmc_fps.json
