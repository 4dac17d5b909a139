#!/usr/bin/env python3
"""Generate tests/golden/*.json from the CPU oracle (small seeded cases of both paths).

These are regression pins for the oracle itself and ready-made expected outputs for the GPU
tests on a box without the oracle's sources changing under them; they are NOT an independent
pin (DESIGN.md section 8 lists those).  Usage: python tools/make_golden.py
"""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))

import oracle  # noqa: E402
import pocs_amd  # noqa: E402

CASES = [
    dict(name="bundled_K3_N512", W=56, K=3, N=512, seed=0x5EED0001),
    dict(name="bundled_K1_N300", W=56, K=1, N=300, seed=7),
    dict(name="bundled_K8_N1000", W=56, K=8, N=1000, seed=2018),
    dict(name="resampled120_K2_N400", W=120, K=2, N=400, seed=99),
]


def main():
    orc = oracle.Oracle()
    env = pocs_amd.load_env()
    base = pocs_amd.load_plan()
    out_dir = ROOT / "tests" / "golden"
    out_dir.mkdir(exist_ok=True)
    for c in CASES:
        plan = base if c["W"] == 56 else pocs_amd.resample_plan(base, c["W"])
        cfg = orc.config(plan, env, K=c["K"])
        g = orc.run_gmm(cfg, c["seed"], c["N"])
        n_mc, hits, _ = orc.run_mc(cfg, c["seed"], c["N"])
        chain = orc.host_chain(cfg, c["seed"])
        doc = dict(case=c,
                   gmm_probability=float(g["prob"]),
                   gmm_waypoint_probabilities=[float(v) for v in g["probs"]],
                   gmm_counts=g["moments"][:, :, :2].astype(int).tolist(),
                   gmm_moments_last=[[float(v) for v in row] for row in g["moments"][-1]],
                   gmm_weights_last=[float(v) for v in g["states"][-1][:, 12]],
                   mc_collided=int(n_mc), mc_probability=n_mc / c["N"],
                   mc_hits_histogram=np.bincount(hits, minlength=1).tolist(),
                   chain_noisy_first=[float(v) for v in chain["noisy"][0]],
                   chain_mu_last=[float(v) for v in chain["mu"][-1]],
                   chain_cov_last=[float(v) for v in chain["cov"][-1]])
        (out_dir / (c["name"] + ".json")).write_text(json.dumps(doc, indent=1))
        print(c["name"], "gmm", g["prob"], "mc", n_mc / c["N"])


if __name__ == "__main__":
    main()
