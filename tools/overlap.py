#!/usr/bin/env python3
"""How the launches of one kernel lie on the time axis, from a rocprofv3 --kernel-trace csv: per launch shape
the mean duration, and over the whole trace the fraction of the busy span during which 0 / 1 / 2+ launches of
the kernel were running (the sub-batches of a call, POCS_OPT_SUB_BATCHES).
usage: overlap.py <dir or kernel_trace.csv> [kernel substring]"""
import csv
import sys
from pathlib import Path

p = Path(sys.argv[1])
key = sys.argv[2] if len(sys.argv) > 2 else "k_gmm_step"
f = p if p.is_file() else next(p.rglob("*kernel_trace.csv"))
rows = [r for r in csv.DictReader(open(f)) if key in r["Kernel_Name"]]
ev = []
for r in rows:
    ev.append((int(r["Start_Timestamp"]), 1))
    ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
depth, last, span = 0, ev[0][0], {}
# only count inside bursts: gaps longer than 50 us (host between calls) are left out
for t, d in ev:
    if depth > 0 or t - last < 50000:
        span[depth] = span.get(depth, 0) + (t - last)
    depth += d
    last = t
tot = sum(span.values())
print("%d launches; time with 0 / 1 / 2 / 3+ of them running: %s" % (
    len(rows), " / ".join("%.1f %%" % (100.0 * span.get(k, 0) / tot) for k in (0, 1, 2)) + " / %.1f %%" % (100.0 * sum(v for k, v in span.items() if k >= 3) / tot)))
dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
print("mean duration %.1f us, busy span %.2f ms" % (sum(dur) / len(dur) / 1e3, tot / 1e6))
