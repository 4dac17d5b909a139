#!/usr/bin/env python3
"""Add a counter record to profiles/traffic.json from the passes tools/profile_round.sh left in gpurun_out/:
HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (the guide's gfx950 correction: FETCH_SIZE counts half of
a wide streaming read) and SQ_INSTS_VALU per launch, for the launch shape of the timed calls (the most frequent grid).
usage: traffic_from_profile.py TAG KEY NAME PATH EVALS_PER_LAUNCH "what"     (KEY = kernel substring incl. template args)"""
import collections
import csv
import json
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
tag, key, name, path, evals, what = sys.argv[1:7]
numerics = sys.argv[7] if len(sys.argv) > 7 else "v9"


def mean_of(d, counter):
    tot, disp = collections.defaultdict(float), collections.defaultdict(set)
    for f in sorted((ROOT / "gpurun_out" / d).rglob("*counter_collection.csv")):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if key in row["Kernel_Name"] and row["Counter_Name"] == counter:
                    g = int(row["Grid_Size"])
                    tot[g] += float(row["Counter_Value"])
                    disp[g].add(row["Dispatch_Id"])
    g = max(disp, key=lambda k: len(disp[k]))
    return tot[g] / len(disp[g]), len(disp[g]), g


w, nw, g = mean_of("pmc_%s_write" % tag, "WRITE_SIZE")
f, nf, _ = mean_of("pmc_%s_fetch" % tag, "FETCH_SIZE")
v, nv, _ = mean_of("pmc_%s_sq" % tag, "SQ_INSTS_VALU")
mk = re.search(r"k_gmm_step<(\d+)", key)
rec = {"bytes_per_launch": int(round((2 * f + w) * 1024)), "evals_per_launch": int(evals), "path": path, "numerics": numerics,
       "components": int(mk.group(1)) if mk else None,
       "valu_insts_per_launch": v,
       "source": "profiles/%s_pmc.txt: (2*FETCH_SIZE + WRITE_SIZE)*1024 B and SQ_INSTS_VALU per launch (%s; grid %d threads, %d / %d / %d dispatches)"
                 % (name, what, g, nw, nf, nv)}
tj = ROOT / "profiles" / "traffic.json"
d = json.loads(tj.read_text())
d[name] = rec
tj.write_text(json.dumps(d, indent=1) + "\n")
print(name, rec["bytes_per_launch"] / 1e6, "MB per launch;", rec["bytes_per_launch"] / rec["evals_per_launch"], "B/eval;",
      v / (int(evals) / 64.0), "VALU instructions per evaluation")
