#!/usr/bin/env python3
"""Per-kernel averages of the counters of one rocprofv3 --pmc CSV directory:
  python tools/pmc_counters.py <pmc_dir> [kernel-name substring[,substring...]] > profiles/xxx.json
"""
import csv
import glob
import json
import os
import sys


def main():
    d = sys.argv[1]
    subs = (sys.argv[2] if len(sys.argv) > 2 else "").split(",")
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = r["Kernel_Name"].replace("void ", "").replace("prcnn::", "").split("(")[0]
                if not any(sub in k for sub in subs) or "at::" in k:
                    continue
                a = acc.setdefault(k, {}).setdefault(r["Counter_Name"], [0, 0.0])
                a[0] += 1
                a[1] += float(r["Counter_Value"])
    out = {k: dict({c: v / n for c, (n, v) in cs.items()}, launches=max(n for n, _ in cs.values()))
           for k, cs in acc.items()}
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
