#!/usr/bin/env python3
"""Soak test of the in-launch hand-off (rows of the virtual slices -> ticket -> last block -> mixture advance):
many runs at varied, uneven sizes and batches, each repeated with the same seed -- every repeat must be
bitwise identical (a stale or torn row would change the sums) -- and, the sums being defined on a run's virtual
slices and not on the launch, identical to the same runs issued one per call."""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import pocs_amd

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(7)
plan, env = pocs_amd.load_plan(), pocs_amd.load_env()
t0, runs, bad = time.time(), 0, 0
t_said = t0
with pocs_amd.Context(0) as c:
    while time.time() - t0 < budget:
        K = int(rng.integers(1, 9))
        N = int(rng.choice([513, 7777, 65537, 262145, 1000003, 3000001]))
        R = int(rng.choice([1, 2, 3, 5, 8, 16, 33, 64]))
        if N * R > 70_000_000:
            R = 1
        seed = int(rng.integers(0, 2 ** 62))
        c.configure(plan, env, K=K, N=N, seed=seed)
        c.set_batch(R)
        c.set_option(pocs_amd.OPT_USE_GRAPH, int(rng.integers(0, 2)))
        ref = None
        for rep in range(3):
            c.set_seed(seed)
            c.run_gmm_estimation()
            got = (tuple(c.batch_probabilities()), c.moments(55, K).tobytes(), c.moments(17, K).tobytes())
            if ref is None:
                ref = got
            elif got != ref:
                bad += 1
                print("MISMATCH K=%d N=%d R=%d seed=%d rep=%d" % (K, N, R, seed, rep))
            runs += R
        c.set_batch(1)
        c.set_seed(seed)
        c.run_gmm_estimation()                      # run 0 on its own launch: the same bits as run 0 of the batch
        if time.time() - t_said > 60:                # (a silent job is taken for a hung one)
            t_said = time.time()
            print("... %d runs, %.0f s, %d mismatches" % (runs, t_said - t0, bad), flush=True)
        if (c.batch_probabilities()[0],) != ref[0][:1]:
            bad += 1
            print("MISMATCH batch vs single K=%d N=%d R=%d seed=%d" % (K, N, R, seed))
print("soak: %d runs in %.0f s, %d mismatches" % (runs, time.time() - t0, bad))
sys.exit(1 if bad else 0)
