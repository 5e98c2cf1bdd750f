#!/bin/bash
# rocprofv3 kernel-trace summary of one probe script: tools/prof_one.sh <script.py> <name>; prints the top kernels
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${2:-one}
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/$1 > $O/out.txt 2> $O/err.txt
cp $O/trace/*/*_kernel_stats.csv $O/kernel_stats.csv
rm -rf $O/trace
python3 - "$O/kernel_stats.csv" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:int(20)]:
    print("%-90s %5s %10.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
