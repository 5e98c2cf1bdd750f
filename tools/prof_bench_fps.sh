#!/bin/bash
# Kernel-trace summary of the pipelined bench: how long the sampling launches take WITH the feature
# kernels in flight beside them.  tools/prof_bench_fps.sh <name> (environment selects the sampler)
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-bench_fps}
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 96 --warmup 8 --no-cpu-baseline --no-extras > $O/out.txt 2> $O/err.txt
cp $O/trace/*/*_kernel_stats.csv $O/kernel_stats.csv
rm -rf $O/trace
python3 - "$O/kernel_stats.csv" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("fps", "fb_s", "gather_point")):
        print("%-70s %5s avg %10.1f us  total %10.1f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
python3 -c "
import json; d = json.load(open('$O/out.txt')); print(d['value'], d['ms_per_step'])"
