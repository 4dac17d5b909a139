#!/usr/bin/env python3
"""GPU check of the summation tree (numerics v7 and later): moments of every waypoint against the oracle's tree order --
bit for bit -- and a run's bits at 1 / 20 / 64 runs per launch and under run-ahead.  usage: tree_check.py [N]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle"))
import oracle
import pocs_amd

SEED = 0x5EED0001
WEYL = 0x9E3779B97F4A7C15
plan, env = pocs_amd.load_plan(), pocs_amd.load_env()
orc = oracle.Oracle()
bad = 0


def runs_of(c, K, R, W):
    out = []
    for r in range(R):
        c.select_batch_run(r)
        out.append((np.array([c.moments(w, K) for w in range(W)]), c.waypoint_probabilities().copy(), c.gmm_state_raw(W - 1, K).copy()))
    return out


with pocs_amd.Context(0) as c:
    for K, N in ((3, 1000), (1, 4097), (3, 20000), (8, 60001), (2, 300000)):
        cfg = orc.config(plan, env, K=K)
        c.configure(plan, env, K=K, N=N, seed=SEED)
        p = c.run_gmm_estimation()
        want = orc.run_gmm(cfg, SEED, N)
        got_m = np.array([c.moments(w, K) for w in range(cfg.W)])
        ok = np.array_equal(got_m, want["moments"]) and p == want["prob"] and np.array_equal(c.waypoint_probabilities(), want["probs"])
        got_s = np.array([c.gmm_state_raw(w, K) for w in range(cfg.W)])
        oks = np.array_equal(got_s[..., :14], want["states"][..., :14])
        print("K=%d N=%d: free-running moments %s, states %s  (max rel moment diff %.3g)" % (
            K, N, "BITWISE" if ok else "DIFFER", "BITWISE" if oks else "DIFFER",
            np.max(np.abs(got_m - want["moments"]) / np.maximum(np.abs(want["moments"]), 1e-300))), flush=True)
        bad += (not ok) + (not oks)
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    K = 3
    cfg = orc.config(plan, env, K=K)
    c.configure(plan, env, K=K, N=N, seed=SEED)
    c.set_batch(1)
    single = []
    for r in range(3):
        c.run_gmm_estimation()
        single.append(runs_of(c, K, 1, cfg.W)[0])
    for R in (20, 64, 7):
        c.set_seed(SEED)
        c.set_batch(R)
        c.run_gmm_estimation()
        got = runs_of(c, K, R, cfg.W)
        for r in range(3):
            same = all(np.array_equal(a, b) for a, b in zip(got[r], single[r]))
            print("N=%d batch %d run %d vs single run: %s" % (N, R, r, "BITWISE" if same else "DIFFER"), flush=True)
            bad += not same
        if R == 20:
            batch20 = got
    c.set_batch(1)
    c.set_option(pocs_amd.OPT_RUN_AHEAD, 64)
    c.set_seed(SEED)
    for r in range(20):
        c.run_gmm_estimation()
        now = (np.array([c.moments(w, K) for w in range(cfg.W)]), c.waypoint_probabilities().copy(), c.gmm_state_raw(cfg.W - 1, K).copy())
        same = all(np.array_equal(a, b) for a, b in zip(now, batch20[r]))
        if r in (0, 1, 19) or not same:
            print("N=%d run-ahead 64, call %d vs run %d of a batch of 20: %s" % (N, r, r, "BITWISE" if same else "DIFFER"), flush=True)
        bad += not same
    c.set_option(pocs_amd.OPT_RUN_AHEAD, 1)
    t0 = time.time()
    for r in (0, 19):
        want = orc.run_gmm(cfg, (SEED + r * WEYL) & (2**64 - 1), N)
        ok = np.array_equal(batch20[r][0], want["moments"]) and np.array_equal(batch20[r][1], want["probs"])
        print("N=%d run %d of the batch of 20 vs the oracle on its seed: %s (%.1f s of oracle)" % (N, r, "BITWISE" if ok else "DIFFER", time.time() - t0), flush=True)
        bad += not ok
print("tree_check:", "OK" if bad == 0 else "%d MISMATCHES" % bad)
sys.exit(1 if bad else 0)
