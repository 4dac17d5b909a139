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
    # BASELINE.json configs[0] exactly: bundled plan, 1k particles, 3-component mixture
    dict(name="cfg1_bundled_K3_N1000", W=56, K=3, N=1000, seed=0x5EED0001),
    # configs[4] at full size: MC roll-outs, 10^5 particles x 500 waypoints (~5 s of oracle)
    dict(name="cfg5_mc_W500_N100000", W=500, K=1, N=100000, seed=0x5EED0001, paths=["mc"]),
    # the reference's second scene (pr2custom.env.xml:58-238: boxes turned +-60 / 90 degrees), obstacle
    # table from tests/golden/pr2custom_env.txt; the first 21 waypoints of the bundled plan (it then
    # runs into the spikes for good) and a slimmer footprint, so that both outcomes occur
    dict(name="pr2custom_K3_N2000", W=21, sub=True, K=3, N=2000, seed=31, env="pr2custom", footprint=[0.0, 0.0, 0.12, 0.10]),
]


def case_plan(c, base):
    if c.get("sub"):                         # the first W waypoints of the bundled plan
        return dict(traj=base["traj"][:c["W"]], odom=base["odom"][:c["W"] - 1])
    return base if c["W"] == 56 else pocs_amd.resample_plan(base, c["W"])


def case_env(c, default_env):
    """The collision world of a case: the bundled pr2test2 table, or a committed scene table."""
    if "env" not in c:
        return default_env
    e = pocs_amd.load_env(ROOT / "tests" / "golden" / (c["env"] + "_env.txt"))
    if "footprint" in c:
        e = dict(e, footprint=list(c["footprint"]))
    return e


def main():
    orc = oracle.Oracle()
    env = pocs_amd.load_env()
    base = pocs_amd.load_plan()
    out_dir = ROOT / "tests" / "golden"
    out_dir.mkdir(exist_ok=True)
    for c in CASES:
        plan = case_plan(c, base)
        cenv = case_env(c, env)
        cfg = orc.config(plan, cenv, K=c["K"])
        paths = c.get("paths", ["gmm", "mc"])
        chain = orc.host_chain(cfg, c["seed"])
        doc = dict(case=c,
                   chain_noisy_first=[float(v) for v in chain["noisy"][0]],
                   chain_mu_last=[float(v) for v in chain["mu"][-1]],
                   chain_cov_last=[float(v) for v in chain["cov"][-1]])
        if "gmm" in paths:
            g = orc.run_gmm(cfg, c["seed"], c["N"])
            doc.update(gmm_probability=float(g["prob"]),
                       gmm_waypoint_probabilities=[float(v) for v in g["probs"]],
                       gmm_counts=g["moments"][:, :, :2].astype(int).tolist(),
                       gmm_moments_last=[[float(v) for v in row] for row in g["moments"][-1]],
                       gmm_weights_last=[float(v) for v in g["states"][-1][:, 12]])
        if "mc" in paths:
            n_mc, hits, parts = orc.run_mc(cfg, c["seed"], c["N"], want_particles=True)
            doc.update(mc_collided=int(n_mc), mc_probability=n_mc / c["N"],
                       mc_hits_histogram=np.bincount(hits, minlength=1).tolist(),
                       # final particles, summed per coordinate in index order (a checksum, exact in the
                       # sense that both sides add the same doubles in the same order on the host)
                       mc_particles_checksum=[float(np.cumsum(parts[:, j])[-1]) for j in range(3)])
        (out_dir / (c["name"] + ".json")).write_text(json.dumps(doc, indent=1))
        print(c["name"], "gmm", doc.get("gmm_probability"), "mc", doc.get("mc_probability"))


if __name__ == "__main__":
    main()
