#!/usr/bin/env python3
"""Tuning aid: does a second context driven from a second host thread hide k_gmm_step's serial
tail (last-block reduce + mixture advance) and the launch gaps?  Prints evals/s for 1 and 2 engines."""
import sys, time, threading
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import pocs_amd

plan, env = pocs_amd.load_plan(), pocs_amd.load_env()
N, K, W = 1000000, 3, 56
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 4


def make(seed):
    c = pocs_amd.Context(0)
    c.configure(plan, env, K=K, N=N, seed=seed)
    c.set_batch(batch)
    return c


def drive(c, n):
    for _ in range(n):
        c.run_gmm_estimation()


for engines in (1, 2, 3):
    ctxs = [make(1 + e) for e in range(engines)]
    for c in ctxs:
        drive(c, 2)
    t0 = time.perf_counter()
    th = [threading.Thread(target=drive, args=(c, calls)) for c in ctxs]
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    evals = engines * calls * batch * N * W
    print("engines %d  batch %d: %.3e evals/s  (%.3f ms per run)" % (engines, batch, evals / dt, 1e3 * dt / (engines * calls * batch)), flush=True)
    for c in ctxs: c.close()
