#!/bin/bash
# tools/pmc_clock.sh <script.py> <name>: kernel trace + (GRBM_GUI_ACTIVE, SQ_INSTS_MFMA, SQ_BUSY_CYCLES) pass
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${2:-clock}
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/$1 > /dev/null 2> $O/trace.err
cp $O/trace/*/*_kernel_stats.csv $O/kernel_stats.csv; rm -rf $O/trace
timeout -s KILL 90 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $O/pmc -- python3 $R/$1 > /dev/null 2> $O/pmc.err
python3 $R/tools/pmc_counters.py $O/pmc mlp_kernel > $O/pmc.json; rm -rf $O/pmc
python3 - $O <<'PY'
import csv, json, sys
o = sys.argv[1]
dur = {r["Name"].replace("void ", "").replace("prcnn::", "").split("(")[0]: float(r["AverageNs"]) for r in csv.DictReader(open(o + "/kernel_stats.csv"))}
for k, c in json.load(open(o + "/pmc.json")).items():
    ns = dur.get(k)
    if not ns: continue
    clk = c["GRBM_GUI_ACTIVE"]
    print("%-50s %8.1f us  %9.0f clocks  %.2f GHz  mfma issue share %.3f" % (k, ns / 1e3, clk, clk / ns, c["SQ_INSTS_MFMA"] * 64 / (1024 * clk)))
PY
