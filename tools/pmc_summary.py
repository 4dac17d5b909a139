#!/usr/bin/env python3
"""Mean per dispatch of every counter in rocprofv3 --pmc output (counter_collection.csv files),
per kernel.  usage: pmc_summary.py dir [dir ...] [--kernel substring]"""
import collections
import csv
import sys
from pathlib import Path

args = [a for a in sys.argv[1:] if not a.startswith("--")]
key = "k_gmm_step"
if "--kernel" in sys.argv:
    key = sys.argv[sys.argv.index("--kernel") + 1]
    args.remove(key)
for d in args:
    for f in sorted(Path(d).rglob("*counter_collection.csv")):
        tot = collections.defaultdict(float)
        disp = collections.defaultdict(set)
        name = None
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if key not in row["Kernel_Name"]:
                    continue
                name = row["Kernel_Name"]
                tot[row["Counter_Name"]] += float(row["Counter_Value"])
                disp[row["Counter_Name"]].add(row["Dispatch_Id"])
        if name is None:
            continue
        n = max(len(v) for v in disp.values())
        print("%s dispatches=%d   (%s)" % (name, n, d))
        for c in sorted(tot):
            print("   %-24s %.1f" % (c, tot[c] / len(disp[c])))
