#!/usr/bin/env python3
"""Mean per dispatch of every counter in rocprofv3 --pmc output (counter_collection.csv files), and of
the kernel durations in --kernel-trace output (kernel_trace.csv), per kernel AND per launch shape (grid
size): one bench run holds launches of several batch sizes (the timed calls, the one-run-per-call
measurement), which must not be averaged together.
usage: pmc_summary.py dir [dir ...] [--kernel substring]"""
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
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if key not in row["Kernel_Name"]:
                    continue
                g = (row["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0], int(row["Grid_Size"]))
                tot[g + (row["Counter_Name"],)] += float(row["Counter_Value"])
                disp[g + (row["Counter_Name"],)].add(row["Dispatch_Id"])
        for g in sorted({k[:2] for k in tot}):
            n = max(len(v) for k, v in disp.items() if k[:2] == g)
            print("%s grid %d threads: %d dispatches   (%s)" % (g[0], g[1], n, d))
            for k in sorted(tot):
                if k[:2] == g:
                    print("   %-24s %.1f" % (k[2], tot[k] / len(disp[k])))
    for f in sorted(Path(d).rglob("*kernel_trace.csv")):
        dur = collections.defaultdict(list)
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if key not in row["Kernel_Name"]:
                    continue
                g = (row["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0],
                     int(row["Grid_Size_X"]) * int(row["Grid_Size_Y"]))
                dur[g].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
        for g in sorted(dur):
            v = dur[g]
            print("%s grid %d threads: %d dispatches, mean %.1f us, min %.1f, max %.1f   (%s)" % (
                g[0], g[1], len(v), sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3, d))
