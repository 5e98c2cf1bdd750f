#!/bin/bash
# Full kernel trace (timestamps, queues) of a short pipelined bench run: tools/trace_bench.sh <name>
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-trace}
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --steps 32 --warmup 8 --no-cpu-baseline --no-extras > $O/out.txt 2> $O/err.txt
cp $O/trace/*/*_kernel_trace.csv $O/kernel_trace.csv
rm -rf $O/trace
python3 - "$O/kernel_trace.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(len(rows), "kernels; columns:", list(rows[0].keys()))
PY
gzip -f $O/kernel_trace.csv
ls -la $O
