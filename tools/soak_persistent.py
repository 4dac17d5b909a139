#!/usr/bin/env python3
"""Soak of the queue-driven kernel (k_gmm_run, POCS_OPT_PERSISTENT) against one launch per waypoint:
the shapes of tests/test_gpu_parity.py::test_persistent_kernel_under_uneven_load, repeated; every
repetition must be bitwise the per-waypoint result.  usage: soak_persistent.py [seconds]"""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import numpy as np
import pocs_amd

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
plan, env = pocs_amd.load_plan(), pocs_amd.load_env()
shapes = [(3, 300000, 7), (8, 150000, 3), (1, 2_000_000, 1), (2, 40001, 20), (3, 20001, 64), (2, 513, 33)]
t0, reps, bad = time.time(), 0, 0
with pocs_amd.Context(0) as c:
    while time.time() - t0 < budget:
        for K, N, R in shapes:
            seed = 1000 + reps
            c.configure(plan, env, K=K, N=N, seed=seed)
            c.set_batch(R)
            c.set_option(pocs_amd.OPT_PERSISTENT, 0)
            c.run_gmm_estimation()
            want_p = list(c.batch_probabilities())
            want_m = np.array([c.moments(w, K) for w in range(56)])
            want_x, want_f = c.gmm_samples(N)
            c.set_option(pocs_amd.OPT_PERSISTENT, 1)
            for rep in range(3):
                c.set_seed(seed)
                c.run_gmm_estimation()
                x, f = c.gmm_samples(N)
                ok = (list(c.batch_probabilities()) == want_p and np.array_equal(np.array([c.moments(w, K) for w in range(56)]), want_m)
                      and np.array_equal(f, want_f) and np.array_equal(x, want_x))
                reps += 1
                if not ok:
                    bad += 1
                    print("MISMATCH K=%d N=%d R=%d seed=%d rep=%d: probs %s moments %s flags %s samples %s" % (
                        K, N, R, seed, rep, list(c.batch_probabilities()) == want_p,
                        np.array_equal(np.array([c.moments(w, K) for w in range(56)]), want_m), np.array_equal(f, want_f), np.array_equal(x, want_x)), flush=True)
print("soak_persistent: %d repetitions in %.0f s, %d mismatches" % (reps, time.time() - t0, bad))
