#!/usr/bin/env python3
"""Tuning aid: split k_gmm_step's launch time into a fixed part (head + tail + mixture advance)
and a per-sample part, by timing the same plan at several sample counts with HIP events
(POCS_OPT_PROFILE) and fitting a line.  usage: fixed_cost.py [K]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import pocs_amd

K = int(sys.argv[1]) if len(sys.argv) > 1 else 3
plan, env = pocs_amd.load_plan(), pocs_amd.load_env()
for batch in (1, 16):
    rows = []
    for n in (2, 65536, 262144, 1000000):
        with pocs_amd.Context(0) as ctx:
            ctx.configure(plan, env, K=K, N=n, seed=1)
            ctx.set_batch(batch)
            ctx.set_option(4, 1)
            best = 1e9
            for rep in range(4):
                ctx.run_gmm_estimation()
                ms, launches = ctx.kernel_time()
                best = min(best, 1e3 * ms / launches)
            rows.append((n * batch, best))
            print("batch %2d  N %8d  k_gmm_step %.2f us/launch" % (batch, n, best), flush=True)
    x = np.array([r[0] for r in rows], float); y = np.array([r[1] for r in rows])
    A = np.vstack([np.ones_like(x), x]).T
    c, res, *_ = np.linalg.lstsq(A, y, rcond=None)
    print("batch %2d  fit: fixed %.2f us + %.3f us per 1e6 samples" % (batch, c[0], c[1] * 1e6), flush=True)
