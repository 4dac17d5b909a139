#!/usr/bin/env python3
"""Per-launch durations and the gaps between consecutive launches of one kernel, from a rocprofv3
--kernel-trace csv (tuning aid).  usage: launch_gaps.py <dir or kernel_trace.csv> [kernel substring]"""
import csv
import sys
from pathlib import Path

p = Path(sys.argv[1])
key = sys.argv[2] if len(sys.argv) > 2 else "k_gmm_step"
f = p if p.is_file() else next(p.rglob("*kernel_trace.csv"))
rows = list(csv.DictReader(open(f)))
ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size_X", r.get("Grid_Size", 0))), int(r.get("Grid_Size_Y", 1))) for r in rows), key=lambda t: t[0])
shapes = {}
for i, (s, e, n, gx, gy) in enumerate(ks):
    if key not in n:
        continue
    d = shapes.setdefault((gx, gy), dict(dur=[], gap=[]))
    d["dur"].append(e - s)
    if i + 1 < len(ks) and key in ks[i + 1][2] and (ks[i + 1][3], ks[i + 1][4]) == (gx, gy):
        d["gap"].append(ks[i + 1][0] - e)
for (gx, gy), d in sorted(shapes.items()):
    dur, gap = sorted(d["dur"]), sorted(d["gap"])
    if not gap:
        continue
    print("grid %6d x %3d: %5d launches  duration mean %.1f us (median %.1f)  gap to the next: mean %.2f us, median %.2f, p90 %.2f"
          % (gx, gy, len(dur), sum(dur) / len(dur) / 1e3, dur[len(dur) // 2] / 1e3, sum(gap) / len(gap) / 1e3, gap[len(gap) // 2] / 1e3, gap[int(0.9 * len(gap))] / 1e3))
