#!/usr/bin/env python3
"""Tuning aid: run the GMM path through the per-waypoint step API (separate advance launch) so a
rocprofv3 --kernel-trace shows k_gmm_advance and k_gmm_step on their own."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import pocs_amd

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 3
plan, env = pocs_amd.load_plan(), pocs_amd.load_env()
with pocs_amd.Context(0) as ctx:
    ctx.configure(plan, env, K=K, N=n, seed=1)
    for rep in range(5):
        ctx.gmm_begin()
        for w in range(56):
            ctx.gmm_step_local(w)
        p = ctx.gmm_end()
    print("prob", p)
