#!/bin/bash
# rocprofv3 kernel-trace summary of ONE eager single-stream pass (tools/step_table.py) and of the
# default bench command; results under gpurun_out/$1 (copy what is to be judged into profiles/).
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-prof}
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_step -- python3 $R/tools/step_table.py > $O/step_table.txt 2> $O/step_table.err
cp $O/trace_step/*/*_kernel_stats.csv $O/step_kernel_stats.csv
rm -rf $O/trace_step
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err
cp $O/trace_bench/*/*_kernel_stats.csv $O/bench_kernel_stats.csv
rm -rf $O/trace_bench
head -40 $O/step_kernel_stats.csv | cut -c1-200
