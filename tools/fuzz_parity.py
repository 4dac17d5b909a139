#!/usr/bin/env python3
"""Extended differential run of the HIP path against the CPU oracle: random worlds (rotated boxes,
off-centre footprints, noise levels, landmark sets, sub-plans, K, N, seeds, batches), more and larger
than tests/test_gpu_parity.py::test_randomised_configurations_match_oracle affords in the suite.
Checks per case, ALL exact (since numerics v7: the oracle restates the summation tree): moments, mixture states and
probabilities of every waypoint, the last waypoint's flags, of run 0 and of the last run of a random batch;
hit counters of the MC path.  usage: fuzz_parity.py [cases] [max_N] [seed]"""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "oracle"))
import numpy as np
import oracle
import pocs_amd

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
max_n = int(sys.argv[2]) if len(sys.argv) > 2 else 40000
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 777)
orc = oracle.Oracle()
plan = pocs_amd.load_plan()
t0, bad = time.time(), 0
with pocs_amd.Context(0) as ctx:
    for case in range(cases):
        M = int(rng.integers(0, 14))
        boxes = np.column_stack([rng.uniform(-3.8, 3.8, M), rng.uniform(-1.8, 1.8, M), rng.uniform(0.03, 0.6, M),
                                 rng.uniform(0.03, 0.6, M), rng.choice([0.0, 0.0, 1.5707963267948966, 1.0, -0.6, 2.2, 3.0], M)])
        fp = [0.0, 0.0, float(rng.uniform(0.1, 0.4)), float(rng.uniform(0.1, 0.4))]
        if case % 3 == 0:
            fp[0], fp[1] = float(rng.uniform(-0.1, 0.1)), float(rng.uniform(-0.1, 0.1))
        env = dict(footprint=fp, boxes=boxes.reshape(-1, 5))
        L = int(rng.integers(1, 12))
        lm = np.vstack([rng.uniform(-4, 4, L), rng.uniform(-2, 2, L)])
        W = int(rng.integers(2, 57))
        start = int(rng.integers(0, 57 - W))
        pl = dict(traj=plan["traj"][start:start + W], odom=plan["odom"][start:start + W - 1])
        K = int(rng.integers(1, 9))
        N = int(rng.integers(50, max_n))
        seed = int(rng.integers(0, 2 ** 62))
        params = dict(pocs_amd.DEFAULTS, landmarks=lm.tolist(), Q=float(rng.uniform(0.01, 0.1)),
                      alphas=[float(a * rng.uniform(0.3, 3)) for a in pocs_amd.DEFAULTS["alphas"]],
                      cov0=(np.eye(3) * rng.uniform(2e-4, 4e-3)).tolist())
        cfg = orc.config(pl, env, K=K, alphas=params["alphas"], Q=params["Q"], landmarks=lm, cov0=params["cov0"])
        ctx.configure(pl, env, params=params, K=K, N=N, seed=seed)
        R = int(rng.choice([1, 1, 2, 5, 17]))
        ctx.set_batch(R)
        ctx.run_gmm_estimation()
        finals = list(ctx.batch_probabilities())
        ok = True
        for r in sorted({0, R - 1}):
            want = orc.run_gmm(cfg, (seed + r * 0x9E3779B97F4A7C15) % 2 ** 64, N, want_samples=True)
            ctx.select_batch_run(r)
            got_m = np.array([ctx.moments(w, K) for w in range(W)])
            got_s = np.array([ctx.gmm_state_raw(w, K) for w in range(W)])[..., :14]
            ok = ok and np.array_equal(got_m, want["moments"]) and np.array_equal(got_s, want["states"][..., :14]) \
                and np.array_equal(ctx.waypoint_probabilities(), want["probs"]) and finals[r] == want["prob"] \
                and np.array_equal(ctx.gmm_samples(N)[1], want["flags"])
        ctx.set_batch(1)
        ctx.set_seed(seed)
        p_mc = ctx.run_simulation()
        n_mc, hits, _ = orc.run_mc(cfg, seed, N)
        ok = ok and p_mc == n_mc / N and np.array_equal(ctx.particles(N)[1], hits)
        if not ok:
            bad += 1
            print("MISMATCH case %d: K=%d N=%d W=%d M=%d seed=%d" % (case, K, N, W, M, seed), flush=True)
        if case % 20 == 19:
            print("%d cases, %d mismatches, %.0f s" % (case + 1, bad, time.time() - t0), flush=True)
print("fuzz: %d cases, %d mismatches, %.0f s" % (cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
